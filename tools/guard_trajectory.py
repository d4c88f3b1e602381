"""How the collapsed discriminator's rounding amplification (DiscriminatorVoxel.guard_collapse) moves while the
synthetic stream trains, and what a step costs on either side of the guard's limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer
dev = torch.device("cuda:0")
batches = [make_batch("dataset", seed=k, ragged="jitter").to(dev) for k in range(8)]
torch.manual_seed(0)
tr = Trainer(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
t0 = time.perf_counter()
for i in range(n):
    b = batches[i % 8]
    b.__dict__.pop("_bg_plan", None)
    build_plan(b)
    st = tr.train_iteration(b, i)
    if i % 100 == 99:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100 * 1e3
        D = tr.discriminator
        ok = D.voxelGNN_D.collapse_ok
        amp = D.guard_collapse(b)
        wn = max(float(p.abs().max()) for p in D.voxelGNN_D.block.parameters())
        print("iter %5d  %.2f ms/step  collapse_ok(before check) %s  amplification %.1f (limit %.1f)  max |block weight| %.3f  d_loss %.4f"
              % (i + 1, dt, ok, amp, D.COLLAPSE_AMPLIFICATION_LIMIT, wn, float(st["d_loss"])), flush=True)
        D.voxelGNN_D.collapse_ok = ok
        t0 = time.perf_counter()
