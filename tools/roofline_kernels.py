"""Launches only the roofline kernels (for rocprofv3 --kernel-trace / --pmc passes): the train step's
dominant GEMM (the generator block's collapsed Linear: [v | mean_j v_j] -> H starting from the position + bias
term, with the per-story term gathered in the epilogue, LeakyReLU and the residual) and the 10^6-edge gather-mean of the microbench."""
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
    batch = make_batch("dataset", seed=0).to(dev)
    plan = build_plan(batch)
    M, H = batch_sizes(batch)["Nv"], 128
    xs = [torch.rand(M, H, device=dev) * 2 - 1, torch.rand(M, H, device=dev) * 2 - 1]
    W = torch.rand(H, 2 * H, device=dev) * 2 - 1
    posb = torch.rand(M, H, device=dev) * 2 - 1
    term = torch.rand(plan.S, H, device=dev)
    for _ in range(iters):
        ops.gemm_raw(xs, W, False, True, act=ops.ACT_LEAKY, slope=0.01, pre_add=term,
                     pre_rows=plan.story_gather.csr.col, residual=xs[0], acc_init=posb)
    # the same product with the story term formed on load (bg_gemm_nbr_f32: what the step launches since round 3)
    pq = torch.rand(plan.S, 2 * H, device=dev)
    for _ in range(iters):
        ops.gemm_nbr_raw(xs, W, pq, plan.ops["story_building_mean"], plan.story_gather.csr.col, act=ops.ACT_LEAKY, slope=0.01,
                         residual=xs[0], acc_init=posb)
if which in ("spmm", "both"):
    batch = make_batch("microbench", seed=0).to(dev)
    plan = build_plan(batch)
    Nv = batch_sizes(batch)["Nv"]
    pairs = 4        # the rotation of bench.py's microbench: no launch finds its operand in the Infinity Cache
    for H in (128, 256):
        xs = [torch.rand(Nv, H, device=dev) * 2 - 1 for _ in range(pairs)]
        outs = [torch.empty(Nv, H, device=dev) for _ in range(pairs)]
        for i in range(max(iters, 2 * pairs)):
            ops.spmm_raw(plan.vox_mean, xs[i % pairs], out=outs[i % pairs])
        del xs, outs
torch.cuda.synchronize()
