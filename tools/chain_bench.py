"""Time the operator-chain kernel variants against separate launches (tuning aid)."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import ops
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
dev = torch.device("cuda:0")
g = make_batch("dataset", seed=0).to(dev)
p = build_plan(g)
b = p.vox_blocks
L = 12
def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
for W in (32, 24, 16):
    Z = torch.rand(p.Nv, (L + 1) * W, device=dev)
    Y = torch.rand(p.Nv, L + 1, W, device=dev)
    for name, blk in (("short6", b), ("short8", dataclasses.replace(b, max_deg=8)),
                      ("lds-index", dataclasses.replace(b, max_deg=1 << 20)), ("separate", None)):
        t1 = timeit(lambda: ops.power_chain_raw(p.vox_mean, Z, L, W, blk))
        t2 = timeit(lambda: ops.horner_chain_raw(p.vox_mean.T, Y, L, W, blk))
        print("W=%2d %-10s powers %7.1f us   horner(A^T) %7.1f us" % (W, name, t1, t2), flush=True)
# where a step's time goes: the same chain keeping only the last step (no per-step global store), and 1 step
W = 32
Z = torch.rand(p.Nv, (L + 1) * W, device=dev)
out = torch.empty(p.Nv, W, device=dev)
for steps in (12, 1):
    t = timeit(lambda: ops._chain_launch(p.vox_mean, b, Z.data_ptr(), Z.stride(0), None, 0, 0, out.data_ptr(), W, 0, steps, W))
    print("W=32 powers, only the last step stored, %2d steps: %7.1f us" % (steps, t), flush=True)
    t = timeit(lambda: ops._chain_launch(p.vox_mean, b, Z.data_ptr(), Z.stride(0), None, 0, 0, Z.data_ptr() + 4 * W, Z.stride(0), W, steps, W))
    print("W=32 powers, every step stored,         %2d steps: %7.1f us" % (steps, t), flush=True)
