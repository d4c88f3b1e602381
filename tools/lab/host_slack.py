"""Is the stream step bound by the host or by the device?  A busy-wait of d microseconds is put at the start of every
step (host only): a device-bound step does not get slower until d exceeds the host's slack.
    python tools/lab/host_slack.py"""
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


def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


K = 200
for rep in range(2):
    for d in (0, 25, 50, 100, 200, 400):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40, 40 + K):
            if d:
                spin(d)
            w.step(i)
        th = time.perf_counter() - t0                      # host done issuing
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        print("delay %4d us: host issue %.3f ms/step, step %.3f ms" % (d, th / K * 1e3, t1 / K * 1e3), flush=True)
