"""Replay A, synchronise, run one eager fill kernel, replay B: which one-call switch makes the result depend on that?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from building_gan_amd import ops, models
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer
cuda = torch.device("cuda:0")
def run(disturb):
    batch = make_batch("small", seed=3, ragged=True).to(cuda)
    torch.manual_seed(0)
    tr = Trainer(cuda)
    tr.capture(batch)
    gA, gB, gC, stats = tr._graphs["with_g"]
    gA.replay()
    if disturb:
        torch.cuda.synchronize()
        xs = [torch.full((1,), float("nan"), device=cuda) for _ in range(24)]
        del xs
    gB.replay(); torch.cuda.synchronize()
    return float(stats["g_adv"]), float(tr.opt_g.grad.double().abs().sum())
print("all on   :", run(False), run(True), run(True), run(False))
