"""10^6-edge gather-mean timing inside a hipGraph of 20 launches (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch, batch_sizes
dev = torch.device("cuda:0")
res = []
for preset in ("microbench", "dataset"):
    batch = make_batch(preset, seed=0).to(dev)
    plan = build_plan(batch)
    Nv = batch_sizes(batch)["Nv"]
    for H in (8, 128, 256):
        x = torch.rand(Nv, H, device=dev)
        out = torch.empty(Nv, H, device=dev)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.spmm_raw(plan.vox_mean, x, out=out)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.spmm_raw(plan.vox_mean, x, out=out)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): g.replay()
        b.record(); torch.cuda.synchronize()
        res.append("%s H=%d: %.1f us" % (preset, H, a.elapsed_time(b) / 200 * 1e3))
print(os.environ.get("BG_LIB_PATH", "default").split("/")[-1], " | ".join(res))
