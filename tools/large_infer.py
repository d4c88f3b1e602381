"""BASELINE configs[4]: large synthetic building (32 stories x 64 x 64 voxels), batched variation sampling
(the reference's generate_multiple_outputs_from_data, util_eval.py:139-162).  Tuning / reporting aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.graph import Batch
from building_gan_amd.models import Generator
from building_gan_amd.synthetic import make_batch, batch_sizes
from building_gan_amd.util_eval import generate_multiple_outputs_from_batch, check_connectivity
from building_gan_amd.plan import build_plan

dev = torch.device("cuda:0")
variations = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
G = Generator(7, 7, 128, 32, 4, 12, dev, max_len=32).to(dev).eval()
one = make_batch("large", seed=0)
batch = Batch.from_data_list([one] * variations) if variations > 1 else one
sz = batch_sizes(batch)
batch.to(dev)
torch.cuda.synchronize()
t = time.perf_counter(); build_plan(batch); torch.cuda.synchronize()
t_plan = time.perf_counter() - t
for _ in range(2):
    res = generate_multiple_outputs_from_batch(batch, variations, G, None, None, None, [dev])
torch.cuda.synchronize()
t = time.perf_counter()
n = 5
for _ in range(n):
    res = generate_multiple_outputs_from_batch(batch, variations, G, None, None, None, [dev])
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
with torch.no_grad():
    pz = torch.rand(sz["Np"], 32, device=dev); vz = torch.rand(sz["Nv"], 32, device=dev)
    out, soft, mask, att, idx = G(batch, pz, vz)
    torch.cuda.synchronize(); t = time.perf_counter()
    inter, missing, gen = check_connectivity(batch, idx, mask["hard"])
    torch.cuda.synchronize(); t_conn = time.perf_counter() - t
print("large x%d: Nv=%d Ev=%d Ec=%d | plan %.1f ms | generate+ratios %.1f ms (%.0f voxels/s, %.1f designs/s) | "
      "check_connectivity %.1f ms | peak mem %.2f GiB" % (variations, sz["Nv"], sz["Ev"], sz["Ec"], t_plan * 1e3, dt * 1e3,
      sz["Nv"] / dt, variations / dt, t_conn * 1e3, torch.cuda.max_memory_allocated() / 2**30))
