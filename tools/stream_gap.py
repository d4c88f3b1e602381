"""Where the stream step's time over the kept-plan iteration goes: the bench's step() against the same iterations with
the plans kept, host-return and GPU-done times of both."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
if os.environ.get('SWITCH_US'): sys.setswitchinterval(float(os.environ['SWITCH_US']) * 1e-6)
args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)
if os.environ.get('H2D_EVENT'):
    torch.cuda.synchronize()
    for b in w.batches:
        ev = torch.cuda.Event(); ev.record(); b.__dict__['_bg_h2d_event'] = ev
from building_gan_amd.plan import build_plan
def run(fn, n=100, warm=20):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(warm + i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3
w.trainer.settle_host() if hasattr(w.trainer, "settle_host") else None
print("stream step (plan of the next batch rebuilt, prefetched): host %.3f ms, done %.3f ms" % run(w.step))
print("   the same, 16 steps queued (host not throttled by a full queue): host %.3f ms, done %.3f ms" % run(w.step, n=16))
def kept(i):
    b = w.batches[i % 8]
    if "_bg_plan" not in b.__dict__: build_plan(b)
    w.trainer.train_iteration(b, i)
for b in w.batches: build_plan(b)
print("same iterations, plans kept:                               host %.3f ms, done %.3f ms" % run(kept))
print("   the same, 16 queued:                                      host %.3f ms, done %.3f ms" % run(kept, n=16))
from building_gan_amd.plan import prefetch_plan
def plan_only(i):
    nxt = w.batches[(i + 1) % 8]
    nxt.__dict__.pop("_bg_plan", None); nxt.__dict__.pop("_bg_plan_pending", None)
    getattr(nxt.voxel_feature, "__dict__", {}).pop("_bg_split", None)
    w.trainer.prefetch(nxt)
    build_plan(nxt)
print("plan alone (prefetch + finish):                            host %.3f ms, done %.3f ms" % run(plan_only))
