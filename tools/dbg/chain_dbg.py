import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from building_gan_amd import ops
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
dev = torch.device("cuda:0")
for preset, ragged, W, L in (("small", True, 8, 5), ("dataset", False, 32, 12)):
    g = make_batch(preset, seed=3, ragged=ragged).to(dev)
    p = build_plan(g)
    b = p.vox_blocks
    print(preset, "blocks", b.n, b.max_rows, b.max_nnz, b.ptr.tolist())
    x = (torch.rand(p.Nv, W) * 2 - 1).to(dev)
    for name, op in (("A", p.vox_mean), ("AT", p.vox_mean.T)):
        Z = torch.zeros(p.Nv, (L + 1) * W + 4, device=dev); Z[:, :W] = x
        Zr = Z.clone()
        ops.power_chain_raw(op, Z, L, W, b)
        ops.power_chain_raw(op, Zr, L, W, None)
        for k in range(L + 1):
            d = (Z[:, k * W:(k + 1) * W] - Zr[:, k * W:(k + 1) * W]).abs()
            print(name, k, float(d.max()), int((d > 0).sum()), float(Zr[:, k * W:(k + 1) * W].abs().max()))
