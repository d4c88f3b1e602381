"""Is the collapsed discriminator's rounding amplification a property of the MONOMIAL basis A^k?  Trains the synthetic
stream until DiscriminatorVoxel.guard_collapse reports a large amplification (or the iteration budget ends), then
trains in the basis ops.D_BASIS (assigned by this tool), and evaluates, in float64, the same bound  max_o max_i (sum_k |Z_k| |W_k|)_io / max_i |v|_io  for
  * the monomial basis   Z_k = A^k U        with the coefficient table the GPU path uses, and
  * the Chebyshev basis  Z_j = T_j(A) U     with the table converted (x^k = sum_j c_kj T_j(x)),
A = the mean-aggregation operator (spectrum in [-1, 1]).  Analysis aid; nothing here is on the product path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from building_gan_amd import models as hm, ops
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
batches = [make_batch("dataset", seed=k, ragged="jitter").to(dev) for k in range(8)]
torch.manual_seed(seed)
tr = Trainer(dev, collapse_check_every=0)
D = tr.discriminator
gnn = D.voxelGNN_D


def analyse(b, tag):
    """Both bases on the CURRENT parameters: the table from the product path (fp32, then float64 copies), the operator
    polynomials applied in float64 here."""
    plan = build_plan(b)
    L = gnn.layer_num
    pos, feat = hm.extract_pos(b.voxel_feature)
    pdm4 = hm._pos_terms(plan, pos)
    K1, dl = L + 1, 8
    src, dst = b.voxel_edge[0], b.voxel_edge[1]
    res = {}
    old = ops.D_BASIS
    try:
        for basis, name in ((0, "monomial"), (1, "Chebyshev")):
            ops.D_BASIS = basis
            with torch.no_grad():
                Zs = gnn._static_powers(plan, feat, pdm4)
                ds = Zs.shape[1] // K1
                W = gnn.coefficients(feat.shape[1], ds, dl).double()
            U = torch.cat([Zs[:, :ds].double(), torch.nn.functional.pad(b.voxel_label, (0, dl - b.voxel_label.shape[1])).double()], 1)
            N = U.shape[0]
            cnt = torch.zeros(N, dtype=torch.float64, device=dev).index_add_(0, dst, torch.ones(src.numel(), dtype=torch.float64, device=dev)).clamp_min(1)
            A = lambda x: torch.zeros_like(x).index_add_(0, dst, x[src]) / cnt[:, None]
            Z = [U, A(U)]
            for k in range(1, L):
                Z.append(2 * A(Z[-1]) - Z[-2] if basis else A(Z[-1]))
            Wk = torch.cat([W[: K1 * ds].view(K1, ds, -1), W[K1 * ds:].view(K1, dl, -1)], dim=1)      # [K1, du, 2H]
            v = sum(Z[k] @ Wk[k] for k in range(K1))
            step = max(1, N // 2048)
            bound = sum(Z[k][::step].abs() @ Wk[k].abs() for k in range(K1))
            amp = float((bound.amax(0) / v[::step].abs().amax(0).clamp_min(1e-30)).amax())
            guard = float(D.guard_collapse(b))
            res[name] = (amp, guard, float(Wk.abs().max()), v)
    finally:
        ops.D_BASIS = old
        D.guard_collapse(b)
    agree = float((res["monomial"][3] - res["Chebyshev"][3]).abs().max() / res["Chebyshev"][3].abs().max())
    print("%s: amplification monomial %.1f (guard, fp32: %.1f) | Chebyshev %.1f (guard %.1f) | max |table| %.2e vs %.2e | "
          "the two forms agree to %.1e" % (tag, res["monomial"][0], res["monomial"][1], res["Chebyshev"][0], res["Chebyshev"][1],
                                         res["monomial"][2], res["Chebyshev"][2], agree), flush=True)


analyse(batches[0], "start      ")
for i in range(budget):
    b = batches[i % 8]
    b.__dict__.pop("_bg_plan", None)
    build_plan(b)
    tr.train_iteration(b, i)
    if i % 200 == 199:
        ok = gnn.collapse_ok
        amp = D.guard_collapse(b)
        gnn.collapse_ok = True
        print("iter %5d amplification %.1f" % (i + 1, amp), flush=True)
        print("        collapse_ok %s (basis %s)" % (ok, "Chebyshev" if ops.D_BASIS else "monomial"), flush=True)
analyse(batches[0], "end (i=%d)" % (i + 1))
