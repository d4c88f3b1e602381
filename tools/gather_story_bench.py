"""gather-mean + story projection: one launch against two, dataset-sized and large batches (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
from building_gan_amd.graph import Batch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, batch in (("dataset", make_batch("dataset", seed=0)), ("large x8", Batch.from_data_list([make_batch("large", seed=0)] * 8))):
    batch.to(dev)
    p = build_plan(batch)
    x = torch.rand(p.Nv, 128, device=dev); W = torch.rand(128, 256, device=dev)
    t1 = timeit(lambda: ops.gather_story_raw(p.vox_mean, p.story_seg, x, W))
    t2 = timeit(lambda: ops.spmm_raw(p.vox_mean, x))
    t3 = timeit(lambda: ops.story_project_raw(p.story_seg, x, W))
    print("%-9s Nv=%d S=%d: one launch %.1f us | gather-mean %.1f us + story projection %.1f us" % (name, p.Nv, p.S, t1, t2, t3), flush=True)
