"""Is the collapsed discriminator's rounding amplification a property of the MONOMIAL basis A^k?  Trains the synthetic
stream until DiscriminatorVoxel.guard_collapse reports a large amplification (or the iteration budget ends), then
evaluates, in float64, the same bound  max_o max_i (sum_k |Z_k| |W_k|)_io / max_i |v|_io  for
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
    plan = build_plan(b)
    L = gnn.layer_num
    pos, feat = hm.extract_pos(b.voxel_feature)
    pdm4 = hm._pos_terms(plan, pos)
    with torch.no_grad():
        Zs = gnn._static_powers(plan, feat, pdm4)
        ds = Zs.shape[1] // (L + 1)
        dl = 8
        W = gnn.coefficients(feat.shape[1], ds, dl).double()
    U = torch.cat([Zs[:, :ds].double(), torch.nn.functional.pad(b.voxel_label, (0, dl - b.voxel_label.shape[1])).double()], 1)
    src, dst = b.voxel_edge[0], b.voxel_edge[1]
    N = U.shape[0]
    cnt = torch.zeros(N, dtype=torch.float64, device=dev).index_add_(0, dst, torch.ones(src.numel(), dtype=torch.float64, device=dev)).clamp_min(1)
    A = lambda x: torch.zeros_like(x).index_add_(0, dst, x[src]) / cnt[:, None]
    Zm = [U]
    for k in range(L):
        Zm.append(A(Zm[-1]))
    Zc = [U, A(U)]
    for k in range(1, L):
        Zc.append(2 * A(Zc[-1]) - Zc[-2])
    K1 = L + 1
    Ws, Wl = W[: K1 * ds].view(K1, ds, -1), W[K1 * ds:].view(K1, dl, -1)
    Wm = torch.cat([Ws, Wl], dim=1)                                   # [K1, du, 2H]
    c = np.zeros((K1, K1))
    for k in range(K1):
        e = np.zeros(K1); e[k] = 1.0
        cc = np.polynomial.chebyshev.poly2cheb(e)
        c[k, : len(cc)] = cc
    Wc = torch.einsum("kj,kuo->juo", torch.from_numpy(c).to(dev), Wm)
    v = sum(Zm[k] @ Wm[k] for k in range(K1))
    v2 = sum(Zc[j] @ Wc[j] for j in range(K1))
    step = max(1, N // 2048)
    bound_m = sum(Zm[k][::step].abs() @ Wm[k].abs() for k in range(K1))
    bound_c = sum(Zc[j][::step].abs() @ Wc[j].abs() for j in range(K1))
    vm = v[::step].abs().amax(0).clamp_min(1e-30)
    print("%s: monomial amplification %.1f | Chebyshev amplification %.1f | both forms agree to %.1e | guard (fp32) %.1f | "
          "max |W_k| monomial %.2e Chebyshev %.2e" % (tag, float((bound_m.amax(0) / vm).amax()), float((bound_c.amax(0) / vm).amax()),
          float((v - v2).abs().max() / v.abs().max()), float(D.guard_collapse(b)), float(Wm.abs().max()), float(Wc.abs().max())), flush=True)


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
        if amp > 400:
            break
analyse(batches[0], "end (i=%d)" % (i + 1))
