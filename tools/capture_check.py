"""Capture + replay of the training iteration on one dataset-sized batch with the one-call nodes switched on / off
(diagnosis): python tools/capture_check.py <head 0|1> <gstack 0|1> [preset]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import models as hm
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer
hm.FUSE_CRITIC_HEAD = sys.argv[1] == "1"
hm.FUSE_GSTACK = sys.argv[2] == "1"
preset = sys.argv[3] if len(sys.argv) > 3 else "dataset"
dev = torch.device("cuda:0")
b = make_batch(preset, seed=0).to(dev)
build_plan(b)
torch.manual_seed(0)
tr = Trainer(dev)
if len(sys.argv) > 4:            # some eager iterations over other batches first, as bench.py does
    others = [make_batch(preset, seed=100 + k, ragged="jitter").to(dev) for k in range(3)]
    for i in range(int(sys.argv[4])):
        o = others[i % 3]
        o.__dict__.pop("_bg_plan", None)
        build_plan(o)
        tr.train_iteration(o, i)
    torch.cuda.synchronize()
    print("eager steps done", flush=True)
print("capturing", sys.argv[1:], flush=True)
tr.capture(b)
print("captured", flush=True)
for i in range(10):
    tr.train_iteration_graphed(i)
torch.cuda.synchronize()
print("replayed ok", flush=True)
