"""Where an eager training step over a stream of distinct batches goes (tuning aid): wall time of the plan build,
of a critic-only iteration and of an iteration with the generator update, then a cProfile of each."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer
dev = torch.device("cuda:0")
batches = [make_batch("dataset", seed=k, ragged="jitter").to(dev) for k in range(8)]
torch.manual_seed(0)
tr = Trainer(dev)
if os.environ.get('BG_OVERLAP') == '1':
    tr._overlap_stream = torch.cuda.Stream()
def plan(b):
    b.__dict__.pop("_bg_plan", None); build_plan(b)
for i in range(10):
    plan(batches[i % 8]); tr.train_iteration(batches[i % 8], i)
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(n): fn(k)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("plan build        %.3f ms" % timeit(lambda k: plan(batches[k % 8])))
print("critic-only iter  %.3f ms" % timeit(lambda k: tr.train_iteration(batches[k % 8], 1)))
print("G-update iter     %.3f ms" % timeit(lambda k: tr.train_iteration(batches[k % 8], 0)))
print("stream step (plan + schedule) %.3f ms" % timeit(lambda k: (plan(batches[k % 8]), tr.train_iteration(batches[k % 8], k)), 40))
what = sys.argv[1] if len(sys.argv) > 1 else "all"
for name, fn in (("plan", lambda k: plan(batches[k % 8])), ("critic", lambda k: tr.train_iteration(batches[k % 8], 1)),
                 ("gupdate", lambda k: tr.train_iteration(batches[k % 8], 0))):
    if what not in ("all", name): continue
    pr = cProfile.Profile(); pr.enable()
    for k in range(10): fn(k)
    torch.cuda.synchronize(); pr.disable()
    print("=" * 30, name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
    pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
