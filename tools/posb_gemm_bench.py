"""Block GEMM of the generator: K = 260 with bias against K = 256 starting from the position + bias term
(acc_init), each inside a hipGraph of 20 launches.  usage: python tools/posb_gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
dev = torch.device("cuda", 0)
Nv, H, S = 11520, 128, 80
torch.manual_seed(1)
v, gm, p4 = (torch.rand(Nv, H, device=dev), torch.rand(Nv, H, device=dev), torch.rand(Nv, 4, device=dev))
W = torch.rand(H, 2 * H + 4, device=dev)
b = torch.rand(H, device=dev)
term = torch.rand(S, H, device=dev)
rows = (torch.arange(Nv, device=dev) // (Nv // S)).to(torch.int32)
Q = ops.gemm_raw([p4], W[:, 2 * H:], False, True, bias=b)


def timed(fn, name):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-40s %.2f us" % (name, e0.elapsed_time(e1) * 1e3 / 200))


timed(lambda: ops.gemm_raw([v, gm, p4], W, False, True, bias=b, act=ops.ACT_LEAKY, slope=0.01, pre_add=term,
                           pre_rows=rows, residual=v), "K=260 + bias")
timed(lambda: ops.gemm_raw([v, gm], W[:, :2 * H], False, True, act=ops.ACT_LEAKY, slope=0.01, pre_add=term,
                           pre_rows=rows, residual=v, acc_init=Q), "K=256 from acc_init")
timed(lambda: ops.gemm_raw([v, gm], W[:, :2 * H], False, True, bias=b, act=ops.ACT_LEAKY, slope=0.01, pre_add=term,
                           pre_rows=rows, residual=v), "K=256 + bias (no position term)")
timed(lambda: ops.gemm_raw([p4], W[:, 2 * H:], False, True, bias=b), "position term [N,4]x[4,H]")
