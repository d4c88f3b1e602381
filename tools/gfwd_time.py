"""Graph-free generator forward (bg_generator_forward_f32) timed as a hipGraph of 10 calls on the dataset batch."""
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
def fn():
    with torch.no_grad():
        tr.generator(batch, pz, vz)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn(); fn()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): fn()
for _ in range(3): g.replay()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): g.replay()
b.record(); torch.cuda.synchronize()
print("generator forward (no autograd graph), Nv = %d: %.1f us per call" % (sz["Nv"], a.elapsed_time(b) / 100 * 1e3))
