"""Split-K weight-gradient product (bg_gemm_tn_splitk_f32 + its reduce) at the step's shapes for several split
targets (tuning aid): time per launch inside a hipGraph of 10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import _lib, ops
from building_gan_amd._lib import check, ptr, stream
dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [("critic label rows  M=104 N=256 R=46080", 104, [256], 46080), ("critic static rows M=416 N=256 R=11520", 416, [256], 11520),
          ("block weight       M=128 N=260 R=11520", 128, [128, 128, 4], 11520), ("story weight       M=128 N=256 R=80", 128, [128, 128], 80),
          ("attention weight   M=132 N=128 R=11520", 132, [128], 11520),
          # the generator update's products over the stacked calls (11 block calls, 6 attention calls)
          ("stacked block weight     M=128 N=260 R=126720", 128, [128, 128, 4], 126720),
          ("stacked attention weight M=132 N=128 R=69120", 132, [128], 69120)]
for name, M, Ns, R in shapes:
    A = torch.rand(R, M, device=dev) - 0.5
    Bs = [torch.rand(R, n, device=dev) - 0.5 for n in Ns]
    N = sum(Ns)
    out = torch.empty(M, N, device=dev)
    tiles = ((M + 63) // 64) * sum((n + 63) // 64 for n in Ns)
    steps = (R + 31) // 32
    res = []
    for target in (512, 1024, 2048, 4096):
        splits = max(1, min(steps, (target + tiles - 1) // tiles, 256))
        partial = torch.empty(splits, M, N, device=dev)
        pb = ops._PtrArr4(*([b.data_ptr() for b in Bs] + [None] * (4 - len(Bs))))
        lb = ops._I64Arr4(*([b.stride(0) for b in Bs] + [0] * (4 - len(Bs))))
        nb = ops._I64Arr4(*(Ns + [0] * (4 - len(Bs))))
        run = lambda: check(lib.bg_gemm_tn_splitk_f32(M, R, ptr(A), M, len(Bs), pb, lb, nb, ptr(out), N, ptr(partial), splits, 0, stream()), "tn")
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): run()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): run()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): g.replay()
        b.record(); torch.cuda.synchronize()
        res.append("target %4d -> %3d splits %6.1f us" % (target, splits, a.elapsed_time(b) / 100 * 1e3))
    print(name, "|", " | ".join(res))
