"""hipGraph-timed GEMM shapes of the train step (20 launches per graph; tuning aid; try BG_GEMM_DMA=0/2/3/4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = lambda *s: torch.rand(*s, device=dev) * 2 - 1
M = 11520
def timeit(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 200 * 1e3
def nt(name, M, N, Ks, pre=False, leaky=False, bias=False, res=False, gather=False, acc=False):
    xs = [R(M, k) for k in Ks]; W = R(N, sum(Ks))
    kw = {}
    if acc: kw["acc_init"] = R(M, N)
    if pre: kw["pre_add"] = R(M, N)
    if gather:
        kw["pre_add"] = R(80, N); kw["pre_rows"] = (torch.arange(M, device=dev) * 80 // M).to(torch.int32)    # rows 0..79 only
    if leaky: kw["act"] = 1
    if bias: kw["bias"] = R(N)
    if res: kw["residual"] = xs[0]
    out = torch.empty(M, N, device=dev)
    us = timeit(lambda: ops.gemm_raw(xs, W, False, True, out=out, **kw))
    fl = 2.0 * M * N * sum(Ks)
    print("%-58s %7.1f us  %6.1f TFLOP/s" % (name, us, fl / us / 1e6), flush=True)
nt("G block K=128+128+4 bias+gather+leaky+res", M, 128, [128, 128, 4], leaky=True, bias=True, res=True, gather=True)
nt("G block K=128+128+4 plain", M, 128, [128, 128, 4])
nt("G block K=128+128 plain", M, 128, [128, 128])
nt("G block K=128+128 bias+gather+leaky+res", M, 128, [128, 128], leaky=True, bias=True, res=True, gather=True)
nt("G block K=128+128 acc_init+gather+leaky+res (the step's)", M, 128, [128, 128], leaky=True, res=True, gather=True, acc=True)
for m in (10130, 11299, 12576):
    nt("same, M=%d" % m, m, 128, [128, 128], leaky=True, res=True, gather=True, acc=True)
nt("NN_v K=128 bias", M, 128, [128], bias=True)
nt("D final N=256 K=416", M, 256, [416])
nt("D label N=256 K=104 pre", M, 256, [104], pre=True)
