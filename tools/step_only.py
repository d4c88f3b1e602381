"""bench.py's timed region alone, for rocprofv3 --kernel-trace: W warm-up steps, a marker launch, K steps of the
stream-of-batches workload (plan build + eager iteration, as bench.py times them), a marker launch.
tools/step_trace_summary.py keeps only the launches between the two markers.

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o s -- python3 tools/step_only.py 40 10
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 10
args = argparse.Namespace(gpus=1, steps=steps, warmup=warm, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)


def marker():
    torch.cumsum(torch.ones(64, device="cuda:0"), 0)          # the only cumsum of the process: the trace's bracket


kind = os.environ.get("BG_STEP_KIND", "stream")      # "critic": critic-only iterations, plans kept (tuning aid)
if kind == "critic":
    step = lambda i: w.trainer.train_iteration(w.batches[i % len(w.batches)], 1)
elif kind == "gupdate":
    step = lambda i: w.trainer.train_iteration(w.batches[i % len(w.batches)], 5)
else:
    step = w.step
for i in range(warm):
    step(i)
w.sync()
marker()
for i in range(warm, warm + steps):
    step(i)
w.sync()
marker()
torch.cuda.synchronize()
print("steps", steps)
