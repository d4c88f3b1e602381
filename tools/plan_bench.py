"""Plan build time (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
dev = torch.device("cuda:0")
for preset in ("dataset", "microbench", "large"):
    b = make_batch(preset, seed=0).to(dev)
    ts = []
    for _ in range(6):
        b.__dict__.pop("_bg_plan", None)
        torch.cuda.synchronize(); t = time.perf_counter()
        build_plan(b)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(preset, "plan build ms:", " ".join("%.2f" % x for x in ts))
