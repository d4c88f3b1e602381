"""Launch the train step's GEMM shapes back to back (for rocprofv3 --kernel-trace; tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = lambda *s: torch.rand(*s, device=dev) * 2 - 1
M = 11520
cases = []
def nt(name, M, N, Ks, **kw):
    xs = [R(M, k) for k in Ks]; W = R(N, sum(Ks))
    extra = {}
    if kw.get("pre"): extra["pre_add"] = R(M, N)
    if kw.get("leaky"): extra["act"] = 1
    if kw.get("bias"): extra["bias"] = R(N)
    cases.append((name, lambda: ops.gemm_raw(xs, W, False, True, **extra)))
def nn(name, M, N, K):
    g = R(M, K); W = R(K, N)
    cases.append((name, lambda: ops.gemm_raw([g], W, False, False)))
nt("G block  NT M=11520 N=128 K=128+128+4 pre+leaky", M, 128, [128, 128, 4], pre=True, leaky=True, bias=True)
nt("NN_v     NT M=11520 N=128 K=128", M, 128, [128], bias=True)
nt("D final  NT M=11520 N=256 K=416", M, 256, [416])
nt("D label  NT M=11520 N=256 K=104 pre", M, 256, [104], pre=True)
nt("D layer  NT M=11520 N=256 K=256+256+4", M, 256, [256, 256, 4], bias=True)
nn("G dx     NN M=11520 N=260 K=128", M, 260, 128)
nn("D dx     NN M=11520 N=104 K=256", M, 104, 256)
nt("coef     NT M=520 N=256 K=256+256", 520, 256, [256, 256])
nn("coef bwd NN M=520 N=256 K=512", 520, 256, 512)
nt("program  NT M=320 N=128 K=128*3", 320, 128, [128, 128, 128], bias=True, leaky=True)
nt("story    NT M=80 N=128 K=128", 80, 128, [128])
for name, fn in cases:
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
print("\n".join(n for n, _ in cases))
