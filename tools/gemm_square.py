import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
dev = torch.device("cuda:0")
def t(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (M, N, K) in [(4096, 4096, 4096), (8192, 256, 512), (65536, 256, 512), (65536, 1024, 512)]:
    A = torch.rand(M, K, device=dev) * 2 - 1
    W = torch.rand(N, K, device=dev) * 2 - 1
    Bn = torch.rand(K, N, device=dev) * 2 - 1
    fl = 2.0 * M * N * K
    us = t(lambda: ops.gemm_raw([A], W, False, True)); r1 = fl / us / 1e6
    us2 = t(lambda: ops.gemm_raw([A], Bn, False, False)); r2 = fl / us2 / 1e6
    print(os.environ.get("BG_GEMM_TILE", "auto"), (M, N, K), "NT %.1f us %.1f TF | NN %.1f us %.1f TF" % (us, r1, us2, r2))
