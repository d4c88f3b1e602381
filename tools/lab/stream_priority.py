"""Does a high-priority main stream keep the prefetched plan's kernels (side stream, normal priority) out of the
iteration's way?  Stream step timed on the default stream and on a priority -1 stream.
    python tools/lab/stream_priority.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0, force_dist=False)
w = bench.HipWorkload(args, 0, 1)
try:
    print("priority range", torch.cuda.Stream.priority_range(), flush=True)
except Exception as e:  # noqa: BLE001
    print("priority_range:", e, flush=True)


def run(K, base):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(base, base + K):
        w.step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for i in range(40):
    w.step(i)
hi = torch.cuda.Stream(priority=-1)
for rep in range(3):
    a = run(200, 40)
    hi.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hi):
        for i in range(40, 60):
            w.step(i)
        b = run(200, 40)
    torch.cuda.current_stream().wait_stream(hi)
    print("default stream %.3f ms/step, priority -1 stream %.3f ms/step" % (a, b), flush=True)
