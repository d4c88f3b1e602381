"""Time the two kinds of training iteration separately (tuning aid).
usage: phase_bench.py [graph|eager] [d|g|both] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
which = sys.argv[2] if len(sys.argv) > 2 else "both"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
batch = make_batch("dataset", seed=0).to(dev)
torch.manual_seed(0)
tr = Trainer(dev)
build_plan(batch)
if mode == "graph":
    tr.capture(batch, overlap=os.environ.get('OVERLAP', '0') == '1')
    fn = tr.train_iteration_graphed
else:
    fn = lambda i: tr.train_iteration(batch, i)
for i in range(5):
    fn(i)
torch.cuda.synchronize()
for name, it in (("d", 1), ("g", 0)):
    if which not in (name, "both"):
        continue
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn(it)
    b.record()
    torch.cuda.synchronize()
    print("%s iteration kind=%s: %.3f ms" % (mode, name, a.elapsed_time(b) / iters))
