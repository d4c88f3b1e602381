"""cProfile of the bench's stream step (host side), sorted by own time."""
import argparse, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0,
                          force_dist=bool(os.environ.get("FORCE_DIST")))     # FORCE_DIST=1: the collectives on a 1-rank RCCL group
w = bench.HipWorkload(args, 0, 1)
for i in range(30): w.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(30, 110): w.step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SORT", "tottime")).print_stats(45)
