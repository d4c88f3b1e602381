"""Launches only the two roofline kernels (for rocprofv3 --kernel-trace / --pmc passes):
the discriminator message GEMM at the train-step shape and the 10^6-edge gather-mean."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch, batch_sizes

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
if which in ("gemm", "both"):
    M, H = 11520, 256
    xs = [torch.rand(M, H, device=dev) * 2 - 1, torch.rand(M, H, device=dev) * 2 - 1, torch.rand(M, 4, device=dev) * 2 - 1]
    W = torch.rand(H, 2 * H + 4, device=dev) * 2 - 1
    b = torch.rand(H, device=dev)
    g = torch.rand(M, H, device=dev) * 2 - 1
    for _ in range(iters):
        ops.gemm_raw(xs, W, False, True, bias=b, residual=xs[0])
        ops.gemm_raw([g], W, False, False)
        ops.gemm_tn_raw(g, xs)
if which in ("spmm", "both"):
    batch = make_batch("microbench", seed=0).to(dev)
    plan = build_plan(batch)
    Nv = batch_sizes(batch)["Nv"]
    for H in (128, 256):
        x = torch.rand(Nv, H, device=dev) * 2 - 1
        for _ in range(iters):
            ops.spmm_raw(plan.vox_mean, x)
torch.cuda.synchronize()
