"""cProfile of plan.prefetch_plan + build_plan on the host (the stream step pays it once per step)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.plan import build_plan, prefetch_plan
from building_gan_amd.synthetic import make_batch
dev = torch.device("cuda:0")
bs = [make_batch("dataset", seed=k, ragged="jitter").to(dev) for k in range(8)]
def once(b):
    b.__dict__.pop("_bg_plan", None)
    prefetch_plan(b)
    return build_plan(b)
for b in bs: once(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(10):
    for b in bs: once(b)
print("prefetch_plan + build_plan: %.3f ms per batch (idle GPU)" % ((time.perf_counter() - t0) / 80 * 1e3))
pr = cProfile.Profile(); pr.enable()
for r in range(10):
    for b in bs: once(b)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
