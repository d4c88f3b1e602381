import os, sys
sys.path.insert(0, "/root/repo")
import torch
from building_gan_amd import ops
dev = torch.device("cuda:0")
R = lambda *s: torch.rand(*s, device=dev) * 2 - 1
M = 1048576
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
xs = [R(M, 128), R(M, 128), R(M, 4)]; W = R(128, 260); b = R(128); term = R(256, 128)
rows = (torch.arange(M, device=dev) // 4096).to(torch.int32)
out = torch.empty(M, 128, device=dev)
us = t(lambda: ops.gemm_raw(xs, W, False, True, bias=b, act=1, pre_add=term, pre_rows=rows, residual=xs[0], out=out))
print("block GEMM M=1M full epilogue: %.0f us %.1f TF/s" % (us, 2.0 * M * 128 * 260 / us / 1e6))
us = t(lambda: ops.gemm_raw(xs, W, False, True, out=out))
print("block GEMM M=1M plain: %.0f us %.1f TF/s" % (us, 2.0 * M * 128 * 260 / us / 1e6))
us = t(lambda: ops.gemm_raw(xs[:1], W[:, :128].contiguous(), False, True, bias=b, out=out))
print("NN_v GEMM M=1M: %.0f us %.1f TF/s" % (us, 2.0 * M * 128 * 128 / us / 1e6))
