"""Unprofiled host time of the pieces of prefetch_plan + build_plan (perf_counter around each, idle GPU)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import plan as P, _lib
from building_gan_amd.synthetic import make_batch
dev = torch.device("cuda:0")
bs = [make_batch("dataset", seed=k, ragged="jitter").to(dev) for k in range(8)]
for b in bs:
    b.__dict__.pop("_bg_plan", None); P.prefetch_plan(b); P.build_plan(b)
torch.cuda.synchronize()
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
N = 20
for r in range(N):
    for g in bs:
        g.__dict__.pop("_bg_plan", None)
        t0 = time.perf_counter(); side = P.prefetch_stream(dev); main = torch.cuda.current_stream(dev); tick("streams", t0)
        t0 = time.perf_counter(); P.order_behind_producer(g, side, main); tick("order_behind_producer", t0)
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            t1 = time.perf_counter()
            pending = P._build_plan_one_call(g, int(g.num_stories), defer=True)
            tick("  _build_plan_one_call(defer)", t1)
        tick("with stream(side) total", t0)
        t0 = time.perf_counter()
        for t in (pending.ai, pending.af): t.record_stream(main)
        tick("record_stream arenas", t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); torch.cuda.current_stream(dev).wait_event(pending.done); tick("wait_event", t0)
        t0 = time.perf_counter(); host = pending.host_buf.tolist(); tick("tolist", t0)
        t0 = time.perf_counter(); P._finish_plan_one_call(g, dev, pending.ai, pending.af, pending.oi, pending.of, pending.sizes, host); tick("_finish_plan_one_call (views)", t0)
for k, v in T.items():
    print("%-36s %7.1f us" % (k, v / (N * 8) * 1e6))
