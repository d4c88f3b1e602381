"""The generator block product (N = 128, K = 128 + 128, the step's epilogue) over the stream's range of M: where the
panel height switches (tuning aid; BG_GEMM_PANEL_T = 2..4 forces it in a tuning build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = lambda *s: torch.rand(*s, device=dev) * 2 - 1
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
for M in [int(x) for x in sys.argv[1:]] or [8192, 10130, 11299, 11520, 12288, 12289, 12576, 14000, 16384]:
    xs = [R(M, 128), R(M, 128)]; W = R(128, 256)
    kw = dict(acc_init=R(M, 128), pre_add=R(80, 128), pre_rows=(torch.arange(M, device=dev) * 80 // M).to(torch.int32), act=1, residual=xs[0])
    out = torch.empty(M, 128, device=dev)
    us = timeit(lambda: ops.gemm_raw(xs, W, False, True, out=out, **kw))
    print("M = %6d : %6.2f us  %5.1f TFLOP/s" % (M, us, 2.0 * M * 128 * 256 / us / 1e6), flush=True)
