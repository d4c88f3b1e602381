"""bg_gemm_nbr_f32 against bg_spmm_add_f32 + bg_gemm_f32 on the generator block's shape (20 launches per hipGraph)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import _lib, ops
lib = _lib.load()
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
M, N, S, B = 11520, 128, 80, 8
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 10             # stories per building
rowptr = (torch.arange(S + 1) * deg).to(torch.int32).to(dev)
col = torch.cat([torch.arange(deg) + (s // deg) * deg for s in range(S)]).to(torch.int32).to(dev)
count = torch.full((S,), float(deg), device=dev)
pq = R(S, 2 * N); rows = (torch.arange(M, device=dev) * S // M).to(torch.int32)
xs = [R(M, 128), R(M, 128)]; W = R(N, 256); acc0 = R(M, N); out = torch.empty(M, N, device=dev); term = torch.empty(S, N, device=dev)
pa, la, ka = (C.c_void_p * 4)(), (C.c_int64 * 4)(), (C.c_int64 * 4)()
for i, x in enumerate(xs): pa[i], la[i], ka[i] = x.data_ptr(), x.stride(0), x.shape[1]
st = lambda: torch.cuda.current_stream().cuda_stream
def fused():
    assert lib.bg_gemm_nbr_f32(M, N, 2, pa, la, ka, W.data_ptr(), 256, out.data_ptr(), N, None, 1, 0.01, xs[0].data_ptr(), 128,
                               pq.data_ptr(), 2 * N, pq.data_ptr() + 4 * N, 2 * N, rowptr.data_ptr(), col.data_ptr(),
                               count.data_ptr(), rows.data_ptr(), acc0.data_ptr(), N, None, 0, st()) == 0
def two():
    assert lib.bg_spmm_add_f32(pq.data_ptr() + 4 * N, 2 * N, rowptr.data_ptr(), col.data_ptr(), count.data_ptr(), 2, None, 0,
                               pq.data_ptr(), 2 * N, None, 0, None, 0, None, None, 0, term.data_ptr(), N, S, N, st()) == 0
    ops.gemm_raw(xs, W, False, True, out=out, act=1, residual=xs[0], pre_add=term, pre_rows=rows, acc_init=acc0)
def plain():
    ops.gemm_raw(xs, W, False, True, out=out, act=1, residual=xs[0], pre_add=term, pre_rows=rows, acc_init=acc0)
print("neighbours per row %d: one launch %.2f us | two launches %.2f us | the product alone %.2f us" % (deg, timeit(fused), timeit(two), timeit(plain)))
