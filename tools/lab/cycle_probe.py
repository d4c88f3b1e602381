"""Which objects of a train step end up in reference cycles (only the cyclic collector frees them: its collections are the
host-side pauses of tools/lab/mode_probe.py)?  20 steps with gc.DEBUG_SAVEALL, then a histogram of gc.garbage by type and,
for the commonest types, what refers to one of them.
    python tools/lab/cycle_probe.py"""
import argparse, collections, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0, force_dist=False)
w = bench.HipWorkload(args, 0, 1)
for i in range(30):
    w.step(i)
torch.cuda.synchronize()
gc.collect()
gc.disable()
gc.set_debug(gc.DEBUG_SAVEALL)
N = int(os.environ.get("PROBE_STEPS", "20"))
for i in range(30, 30 + N):
    w.step(i)
torch.cuda.synchronize()
n = gc.collect()
gc.set_debug(0)
hist = collections.Counter(type(o).__name__ for o in gc.garbage)
print("steps", N, "unreachable", n, "saved", len(gc.garbage), flush=True)
for name, c in hist.most_common(25):
    print("%8d  %s" % (c, name))


def describe(o):
    t = type(o).__name__
    if t == "function":
        return "function %s (%s:%d)" % (o.__qualname__, o.__code__.co_filename.rsplit("/", 1)[-1], o.__code__.co_firstlineno)
    if t == "cell":
        try:
            return "cell -> %s" % type(o.cell_contents).__name__
        except ValueError:
            return "cell (empty)"
    if t == "dict":
        return "dict keys=%s" % list(o.keys())[:8]
    if t in ("tuple", "list"):
        return "%s of %s" % (t, [type(x).__name__ for x in o[:6]])
    return t + " " + repr(o)[:100].replace("\n", " ")


seen = set()
for o in gc.garbage:
    t = type(o).__name__
    if t in seen or t in ("int", "float", "str"):
        continue
    seen.add(t)
    print("--", describe(o))
    for r in gc.get_referrers(o)[:4]:
        if r is gc.garbage:
            continue
        print("     <-", describe(r))
    if len(seen) > 14:
        break
