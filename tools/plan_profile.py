"""cProfile of build_plan (tuning aid)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
dev = torch.device("cuda:0")
b = make_batch("dataset", seed=0).to(dev)
for _ in range(3):
    b.__dict__.pop("_bg_plan", None); build_plan(b)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    b.__dict__.pop("_bg_plan", None); build_plan(b)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
