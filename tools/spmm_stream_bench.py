"""10^6-edge gather-mean with rotating operand / result pairs (HBM streaming, as bench.py's microbench) and with one
pair (Infinity-Cache resident); tuning aid.  BG_SPMM_BATCH=0/1/2/4 selects the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch, batch_sizes
dev = torch.device("cuda:0")
batch = make_batch("microbench", seed=0).to(dev)
plan = build_plan(batch)
Nv = batch_sizes(batch)["Nv"]
res = []
for H in (128, 256):
    for pairs in (4, 1):
        xs = [torch.rand(Nv, H, device=dev) * 2 - 1 for _ in range(pairs)]
        outs = [torch.empty(Nv, H, device=dev) for _ in range(pairs)]
        for i in range(8): ops.spmm_raw(plan.vox_mean, xs[i % pairs], out=outs[i % pairs])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(48): ops.spmm_raw(plan.vox_mean, xs[i % pairs], out=outs[i % pairs])
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 48 * 1e3
        comp = 2 * Nv * 4 * H + 1006896 * 4 + Nv * 8
        res.append("H=%d pairs=%d: %.1f us (%.2f TB/s compulsory)" % (H, pairs, us, comp / us / 1e6))
        ref = ops.spmm_raw(plan.vox_mean, xs[0])
        del xs, outs
print("BG_SPMM_BATCH=%s | " % os.environ.get("BG_SPMM_BATCH", "1") + " | ".join(res))
