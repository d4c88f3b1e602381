"""hipGraph-timed products of the collapsed discriminator's coefficient recurrence (bg_gemm_cheb_f32: [rows, 512 | 512] x
[1024, 512], rows = 80 .. 520) -- forward (B = Wc^T, pitch 4H+8) and backward (B = wab).  20 launches per graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = lambda *s: torch.rand(*s, device=dev) * 2 - 1
H2, du, nb = 256, 40, 13
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
X = R(nb * du + du, H2); E = R(nb * du, H2); C = torch.empty(nb * du + du, H2, device=dev)
Wc = R(H2, 4 * (H2 // 2) + 8); wab = R(2 * H2, H2)
def st(): return torch.cuda.current_stream().cuda_stream
tot = 0.0
for rows in [80, 120, 160, 200, 240, 280, 320, 360, 400, 440, 480, 520]:
    def f():
        rc = lib.bg_gemm_cheb_f32(1, rows, H2, X.data_ptr(), H2, H2, du, nb, 0, Wc.data_ptr(), Wc.stride(0), C.data_ptr(), H2,
                                  X.data_ptr(), H2, E.data_ptr(), H2, st())
        assert rc == 0, rc
    us = timeit(f); tot += us
    print("forward  rows %3d  tiles %3d : %6.2f us" % (rows, (rows + 31) // 32 * (H2 // 32), us), flush=True)
print("forward, 12 layers: %.1f us" % tot)
def b():
    rc = lib.bg_gemm_cheb_f32(0, nb * du, H2, X.data_ptr(), H2, H2, du, nb, 1, wab.data_ptr(), H2, C.data_ptr(), H2,
                              X.data_ptr(), H2, None, 0, st())
    assert rc == 0, rc
print("backward rows 520 (x12 per pass)  : %6.2f us" % timeit(b), flush=True)
# the K <= 256 shapes of the step (program GNN, per-story terms, decoders): bg_gemm_f32 through ops.gemm_raw
from building_gan_amd import ops
for (M, N, Ks, tb) in [(308, 128, [128, 128], True), (308, 128, [128], True), (88, 128, [128], True), (88, 256, [256], True),
                       (520, 256, [256], True), (308, 128, [128, 128], False), (88, 256, [256], False), (308, 128, [96], True)]:
    xs = [R(M, k) for k in Ks]; W = R(N, sum(Ks)) if tb else R(sum(Ks), N)
    out = torch.empty(M, N, device=dev); b = R(N); res = R(M, N)
    us = timeit(lambda: ops.gemm_raw(xs, W, False, tb, out=out, bias=b, act=1, residual=res))
    print("M %4d N %3d K %-12s transB %d : %6.2f us" % (M, N, Ks, tb, us), flush=True)
