"""Host-side profile (cProfile) of one kind of iteration: where the Python time of the eager step goes.
    python tools/host_profile.py gupdate|critic [n]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "gupdate"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)
it = 5 if kind == "gupdate" else 1
for i in range(10):
    w.trainer.train_iteration(w.batches[i % 8], it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    w.trainer.train_iteration(w.batches[i % 8], it)
t_host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print("%s iteration: host returns after %.2f ms, GPU done after %.2f ms (per iteration, %d iterations queued)" % (kind, t_host * 1e3, t_all * 1e3, n))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    w.trainer.train_iteration(w.batches[i % 8], it)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
