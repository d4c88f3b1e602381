"""cProfile cumulative times of the stream step against the same iterations with kept plans: what a fresh plan costs the host."""
import argparse, cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from building_gan_amd.plan import build_plan
args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)
def kept(i):
    b = w.batches[i % 8]
    if "_bg_plan" not in b.__dict__: build_plan(b)
    w.trainer.train_iteration(b, i)
def prof(fn, label):
    for i in range(30): fn(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for i in range(30, 110): fn(i)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr)
    rows = {}
    for (f, ln, nm), (cc, nc, tt, ct, callers) in st.stats.items():
        rows["%s:%d(%s)" % (os.path.basename(f), ln, nm)] = (nc, tt / 80 * 1e3, ct / 80 * 1e3)
    return rows
a = prof(w.step, "stream")
for b in w.batches: build_plan(b)
k = prof(kept, "kept")
keys = sorted(set(a) | set(k), key=lambda x: -(a.get(x, (0, 0, 0))[1] - k.get(x, (0, 0, 0))[1]))
print("%-64s %8s %8s %8s %8s" % ("function (own time, ms per step)", "stream", "kept", "diff", "calls"))
for x in keys[:28]:
    sa, sk = a.get(x, (0, 0, 0)), k.get(x, (0, 0, 0))
    print("%-64s %8.3f %8.3f %8.3f %4d/%-4d" % (x[:64], sa[1], sk[1], sa[1] - sk[1], sa[0] // 80, sk[0] // 80))
print("total own: stream %.3f, kept %.3f" % (sum(v[1] for v in a.values()), sum(v[1] for v in k.values())))
