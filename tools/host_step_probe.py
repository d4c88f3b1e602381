"""Host-side time of the stream step's two parts (train_iteration, prefetch of the next batch's plan), per iteration kind,
and the GPU's idle share: is the eager stream step host- or GPU-bound?"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = argparse.Namespace(gpus=1, steps=60, warmup=10, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)
tr = w.trainer
for i in range(40):
    w.step(i)
w.settle_host()
w.sync()
orig_it, orig_pf = tr.train_iteration, tr.prefetch
acc = {"it_c": [], "it_g": [], "pf": []}
def timed_it(batch, i, *a, **k):
    t = time.perf_counter(); r = orig_it(batch, i, *a, **k); acc["it_g" if i % tr.n_critic_g == 0 else "it_c"].append(time.perf_counter() - t); return r
def timed_pf(batch):
    t = time.perf_counter(); r = orig_pf(batch); acc["pf"].append(time.perf_counter() - t); return r
tr.train_iteration, tr.prefetch = timed_it, timed_pf
from building_gan_amd import plan as P
acc["bp"], acc["sync"] = [], []
orig_bp, orig_fin = w.build_plan, P._PendingPlan.finish
def timed_bp(b):
    t = time.perf_counter(); r = orig_bp(b); acc["bp"].append(time.perf_counter() - t); return r
def timed_fin(self, graph):
    t = time.perf_counter(); self.done.synchronize(); acc["sync"].append(time.perf_counter() - t); return orig_fin(self, graph)
w.build_plan = timed_bp
P._PendingPlan.finish = timed_fin
t0 = time.perf_counter()
n = 200
for i in range(40, 40 + n):
    w.step(i)
t_host = time.perf_counter() - t0
w.sync()
t_all = time.perf_counter() - t0
med = lambda v: sorted(v)[len(v) // 2] * 1e3
print("stream step: %.3f ms/step (host loop done after %.3f ms/step)" % (t_all / n * 1e3, t_host / n * 1e3))
print("host time, median: critic iteration %.3f ms, iteration with G update %.3f ms, prefetch of the next plan %.3f ms (mean %.3f)"
      % (med(acc["it_c"]), med(acc["it_g"]), med(acc["pf"]), sum(acc["pf"]) / len(acc["pf"]) * 1e3))
print("build_plan at the start of a step (finishing the prefetched plan): median %.3f ms, of which waiting for the read-back %.3f ms"
      % (med(acc["bp"]), med(acc["sync"])))
print("host per 5 steps: %.2f ms; wall per 5 steps %.2f ms" % (4 * med(acc["it_c"]) + med(acc["it_g"]) + 5 * med(acc["pf"]), t_all / n * 5e3))
