"""Runs the generator forward a few times (for rocprofv3 --kernel-trace; tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch, batch_sizes
from building_gan_amd.train import Trainer
dev = torch.device("cuda:0")
batch = make_batch("dataset", seed=0).to(dev)
sz = batch_sizes(batch)
torch.manual_seed(0)
tr = Trainer(dev)
build_plan(batch)
pz = torch.rand(sz["Np"], 32, device=dev); vz = torch.rand(sz["Nv"], 32, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "nograd"
for _ in range(3):
    if mode == "nograd":
        with torch.no_grad():
            tr.generator(batch, pz, vz)
    else:
        gen = tr.generator(batch, pz, vz)
        tr._phase_g(batch, (sz["B"], 80, sz["Nv"]), gen)
    torch.cuda.synchronize()
