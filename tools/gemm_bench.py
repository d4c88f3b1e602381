"""GEMM shape sweep on the GPU (tuning aid; not part of the product path)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops

dev = torch.device("cuda:0")

def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

M = int(os.environ.get("M", 11520))
res = {}
for name, (N, Ks) in {"D_msg": (256, [256, 256, 4]), "D_upd": (256, [256, 256]), "G_msg": (128, [128, 128, 4]),
                      "G_upd": (128, [128] * 4), "att": (128, [128])}.items():
    xs = [torch.rand(M, k, device=dev) * 2 - 1 for k in Ks]
    K = sum(Ks)
    W = torch.rand(N, K, device=dev) * 2 - 1
    g = torch.rand(M, N, device=dev) * 2 - 1
    xc = torch.cat(xs, 1)
    fl = 2.0 * M * N * K
    us = t(lambda: ops.gemm_raw(xs, W, False, True))
    res[name + "_fwdNT"] = (round(us, 1), round(fl / us / 1e6, 1))
    us = t(lambda: ops.gemm_raw([g], W, False, False))
    res[name + "_dxNN"] = (round(us, 1), round(fl / us / 1e6, 1))
    us = t(lambda: ops.gemm_tn_raw(g, xc))
    res[name + "_dwTN"] = (round(us, 1), round(fl / us / 1e6, 1))
    us = t(lambda: ops.colsum_raw(g))
    res[name + "_colsum"] = (round(us, 1), 0)
print(os.environ.get("BG_GEMM_TILE", "auto"), json.dumps(res))
