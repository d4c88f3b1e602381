"""Run-to-run spread of the stream step: 30 blocks of 100 steps in one process, host-issue time and time to device idle per
block.  A block whose host-issue time is above the usual device time is host-bound.
    python tools/lab/mode_probe.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0, force_dist=False)
w = bench.HipWorkload(args, 0, 1)
for i in range(60):
    w.step(i)
torch.cuda.synchronize()
import gc
mode = os.environ.get("PROBE_GC", "")
if mode == "off":
    gc.disable()
elif mode == "freeze":
    gc.collect(); gc.freeze()
print("gc", mode or "default", gc.get_threshold(), gc.get_count(), flush=True)
K = 100
rows = []
for blk in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        w.step(60 + i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    rows.append((th / K * 1e3, t1 / K * 1e3))
print(" ".join("%.3f/%.3f" % r for r in rows), flush=True)
print("gc stats", gc.get_stats(), flush=True)
