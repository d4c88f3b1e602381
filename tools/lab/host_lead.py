"""How far does the host get ahead of the device?  After a synchronisation K steps are issued and the host's clock is
read before and after the final synchronisation: host-issue time < step time means the device is the bound.
    python tools/lab/host_lead.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0, force_dist=False)
w = bench.HipWorkload(args, 0, 1)
for i in range(40):
    w.step(i)
torch.cuda.synchronize()
for K in (1, 2, 4, 8, 16, 64):
    hs, ts = [], []
    base = 40
    for rep in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(base, base + K):
            w.step(i)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        hs.append(th / K * 1e3); ts.append(t1 / K * 1e3)
    hs.sort(); ts.sort()
    print("K %3d: host issue median %.3f ms/step (min %.3f), to device idle %.3f ms/step (min %.3f)"
          % (K, hs[len(hs) // 2], hs[0], ts[len(ts) // 2], ts[0]), flush=True)
