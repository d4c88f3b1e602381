"""Host wrappers and autograd Functions over the C-ABI kernels (include/bgan.h).

Every Function's backward is written with other Functions of this file (or stock differentiable
torch glue), so the family is closed under differentiation: the WGAN-GP penalty's double backward
through the discriminator (/root/reference/Model/losses.py:83-84, create_graph=True) works.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import check, ptr, stream
from .plan import Csr, LinOp, SCALE_DIV, SCALE_NONE, iota_i32

ACT_NONE, ACT_LEAKY = 0, 1
# Basis of the collapsed discriminator's polynomial in the aggregation operator A (include/bgan.h BG_CHAIN_*):
# 1 = Chebyshev T_k(A) (default: |T_k| <= 1 on A's spectrum, the coefficient table stays well conditioned while the
# parameters train), 0 = monomials A^k (the first formulation; its rounding amplification grew past
# DiscriminatorVoxel.COLLAPSE_AMPLIFICATION_LIMIT within ~1000 iterations of training: DESIGN.md section 6).
CHAIN_POWER, CHAIN_CHEB, CHAIN_CLENSHAW = 0, 1, 2
# The collapsed discriminator runs in the Chebyshev basis.  The monomial basis (CHAIN_POWER) stays in the kernels for the
# tests and tools/basis_conditioning.py, which assign this attribute to compare the two; it is NOT a product option (no
# environment variable or constructor argument selects it): its amplification passes the guard's limit after ~1,000
# training iterations.
D_BASIS = CHAIN_CHEB
FUSE_COEF_RECURRENCE = True      # PolyCoefD: the coefficient recurrence and its backward as one host call each

_INPUT_GRADS_ONLY = False
_NO_GRAD_CALL = False      # set by linear(): the call was made under torch.no_grad()


@contextlib.contextmanager
def input_grads_only():
    """Linears evaluated inside this context will not produce weight / bias gradients when they
    are differentiated: only the gradient with respect to their inputs (which still depends
    differentiably on the weights).  Used for the gradient-penalty pass, where the discriminator's
    forward graph is differentiated with respect to the sample only (reference losses.py:83-84) and
    the parameter gradients arrive through the double-backward graph; autograd would otherwise
    compute -- and discard -- a weight-gradient GEMM per layer in each of the two grad calls."""
    global _INPUT_GRADS_ONLY
    prev = _INPUT_GRADS_ONLY
    _INPUT_GRADS_ONLY = True
    try:
        yield
    finally:
        _INPUT_GRADS_ONLY = prev


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """Row-major fp32 with unit inner stride (a row pitch is allowed)."""
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() == 1:
        return t if t.stride(0) == 1 or t.numel() <= 1 else t.contiguous()
    if t.stride(-1) != 1 or (t.dim() == 2 and t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        return t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


# ------------------------------------------------------------------------------- raw launchers
_ZEROS = {}


def zeros_view(*shape, device) -> torch.Tensor:
    """A read-only all-zero fp32 tensor of the given shape: a view of one cached buffer per device (no
    fill launch).  For padding blocks inside torch.cat; never write to it."""
    n = 1
    for d in shape:
        n *= d
    buf = _ZEROS.get(device)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(n, 1 << 16), dtype=torch.float32, device=device)
        _ZEROS[device] = buf
    return buf[:n].view(*shape)


_ONES = {}


def ones_view(n: int, device) -> torch.Tensor:
    """A read-only all-one fp32 vector (cached per device, as zeros_view)."""
    buf = _ONES.get(device)
    if buf is None or buf.numel() < n:
        buf = torch.ones(max(n, 16), dtype=torch.float32, device=device)
        _ONES[device] = buf
    return buf[:n]


def _vec_ok(*ts) -> bool:
    """16-byte path: width % 4 == 0, aligned base, pitch % 4 == 0."""
    return all(t.shape[-1] % 4 == 0 and t.data_ptr() % 16 == 0 and _ld(t) % 4 == 0 for t in ts)


def spmm_raw(op: LinOp, x: torch.Tensor, edge_w: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _lib.require_device(x, "spmm")
    squeeze = x.dim() == 1
    x2 = _f32c(x.view(-1, 1) if squeeze else x)
    c: Csr = op.csr
    if x2.shape[0] != c.n_cols:
        raise _lib.BganError(f"spmm: operand has {x2.shape[0]} rows, plan expects {c.n_cols}")
    H = x2.shape[1]
    if out is None:
        out = torch.empty(c.n_rows, H, dtype=torch.float32, device=x.device)
    eperm = None
    if edge_w is not None:
        edge_w = _f32c(edge_w.reshape(-1))
        if edge_w.numel() != c.nnz:
            raise _lib.BganError("spmm: edge weight count does not match the plan")
        eperm = c.perm
    if c.row_per_workgroup and op.col_mode == SCALE_NONE and _vec_ok(x2, out):
        check(lib.bg_segreduce_sum_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), ptr(edge_w), ptr(eperm),
                                       ptr(op.row_scale), op.row_mode, None, 0, 0, None, ptr(out), _ld(out),
                                       c.n_rows, H, stream()), "bg_segreduce_sum_f32")
    elif c.long_rows and op.col_mode == SCALE_NONE:
        partial = torch.empty(c.n_chunks_ub, H, dtype=torch.float32, device=x.device)
        check(lib.bg_segreduce_sum_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), ptr(edge_w), ptr(eperm),
                                       ptr(op.row_scale), op.row_mode, ptr(c.chunk_ptr), c.rows_per_chunk,
                                       c.n_chunks_ub, ptr(partial), ptr(out), _ld(out), c.n_rows, H, stream()),
              "bg_segreduce_sum_f32")
    else:
        check(lib.bg_spmm_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), ptr(edge_w), ptr(eperm),
                              ptr(op.row_scale), op.row_mode, ptr(op.col_scale), op.col_mode,
                              ptr(out), _ld(out), c.n_rows, H, stream()), "bg_spmm_f32")
    return out.view(-1) if squeeze else out


def spmm_add_raw(op: LinOp, x: torch.Tensor, add1=None, add2=None, out=None, gather=None) -> torch.Tensor:
    """out = op(x) + add1 + add2 [+ gs * gadd[rows]] in one pass (short-row plans only).
    ``gather`` = (gadd [R, H], rows int32 [n_rows], scale [R] or None, scale mode)."""
    lib = _lib.load()
    c: Csr = op.csr
    x2 = _f32c(x)
    if x2.shape[0] != c.n_cols:
        raise _lib.BganError("spmm_add: operand rows do not match the plan")
    H = x2.shape[1]
    if out is None:
        out = torch.empty(c.n_rows, H, dtype=torch.float32, device=x.device)
    a1 = _f32c(add1) if add1 is not None else None
    a2 = _f32c(add2) if add2 is not None else None
    ga = grow = gs = None
    gmode = SCALE_NONE
    if gather is not None:
        ga, grow, gs, gmode = gather
        ga = _f32c(ga)
        if (grow.dtype != torch.int32 or grow.numel() != c.n_rows or ga.shape[1] != H
                or (gs is not None and gs.numel() != ga.shape[0])):
            raise _lib.BganError("spmm_add: gathered addend does not fit the plan")
        if gs is None:
            gmode = SCALE_NONE
    check(lib.bg_spmm_add_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), ptr(op.row_scale), op.row_mode,
                              ptr(op.col_scale), op.col_mode, ptr(a1), _ld(a1) if a1 is not None else 0,
                              ptr(a2), _ld(a2) if a2 is not None else 0,
                              ptr(ga), _ld(ga) if ga is not None else 0, ptr(grow), ptr(gs), gmode,
                              ptr(out), _ld(out), c.n_rows, H, stream()),
          "bg_spmm_add_f32")
    return out


def sddmm_raw(op: LinOp, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """out[e] = <a[row(e)], b[col(e)]> in ORIGINAL edge order of the plan."""
    lib = _lib.load()
    c = op.csr
    a2, b2 = _f32c(a.view(a.shape[0], -1)), _f32c(b.view(b.shape[0], -1))
    if a2.shape[0] != c.n_rows or b2.shape[0] != c.n_cols or a2.shape[1] != b2.shape[1]:
        raise _lib.BganError("sddmm: operand shapes do not match the plan")
    out = torch.empty(c.nnz, dtype=torch.float32, device=a.device)
    check(lib.bg_sddmm_f32(ptr(a2), _ld(a2), ptr(b2), _ld(b2), ptr(c.rowptr), ptr(c.col), ptr(c.perm), ptr(out),
                           c.n_rows, a2.shape[1], stream()), "bg_sddmm_f32")
    return out


def segmax_raw(c: Csr, x: torch.Tensor):
    lib = _lib.load()
    x2 = _f32c(x)
    if x2.shape[0] != c.n_cols or c.chunk_ptr is None:
        raise _lib.BganError("segmax: operand rows / chunk table do not match the plan")
    H = x2.shape[1]
    dev = x.device
    out = torch.empty(c.n_rows, H, dtype=torch.float32, device=dev)
    arg = torch.empty(c.n_rows, H, dtype=torch.int32, device=dev)
    if c.row_per_workgroup and _vec_ok(x2):
        check(lib.bg_segmax_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), None, 0, 0, None, None, ptr(out), H,
                                ptr(arg), c.n_rows, H, stream()), "bg_segmax_f32")
        return out, arg
    pv = torch.empty(c.n_chunks_ub, H, dtype=torch.float32, device=dev)
    pa = torch.empty(c.n_chunks_ub, H, dtype=torch.int32, device=dev)
    check(lib.bg_segmax_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), ptr(c.chunk_ptr), c.rows_per_chunk,
                            c.n_chunks_ub, ptr(pv), ptr(pa), ptr(out), H, ptr(arg), c.n_rows, H, stream()),
          "bg_segmax_f32")
    return out, arg


def segsum_acc_raw(c: Csr, x: torch.Tensor, out: torch.Tensor, edge_w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out += the segments' (edge-weighted) sums of x's rows, one launch (bg_segsum_acc_f32: a workgroup per segment)."""
    lib = _lib.load()
    x2 = _f32c(x)
    H = x2.shape[1]
    if tuple(out.shape) != (c.n_rows, H) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise _lib.BganError("segsum_acc: `out` must be an fp32 [n_rows, H] matrix with contiguous rows")
    check(lib.bg_segsum_acc_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), ptr(edge_w), ptr(c.perm) if edge_w is not None else None,
                                ptr(out), out.stride(0), c.n_rows, H, stream()), "bg_segsum_acc_f32")
    return out


def segmax_batch_raw(c: Csr, x: torch.Tensor):
    """Segment max + arg of ``x`` [n_batch, n_cols, H] over the same segments for every batch, one launch
    (bg_segmax_batch_f32: the one-launch form, a workgroup per segment and batch)."""
    lib = _lib.load()
    if x.dim() != 3 or x.shape[1] != c.n_cols or x.dtype != torch.float32 or not x.is_contiguous():
        raise _lib.BganError("segmax_batch: x must be a contiguous float32 [n_batch, n_cols, H]")
    nb, _, H = x.shape
    out = torch.empty(nb, c.n_rows, H, dtype=torch.float32, device=x.device)
    arg = torch.empty(nb, c.n_rows, H, dtype=torch.int32, device=x.device)
    check(lib.bg_segmax_batch_f32(ptr(x), H, c.n_cols * H, ptr(c.rowptr), ptr(c.col), ptr(out), H, c.n_rows * H, ptr(arg),
                                  c.n_rows * H, c.n_rows, H, nb, stream()), "bg_segmax_batch_f32")
    return out, arg


def scatter_arg_batch_raw(g: torch.Tensor, arg: torch.Tensor, n_rows: int, seg_of: torch.Tensor,
                          sum_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scatter_arg_raw's row-lookup form for g, arg [n_batch, n_seg, H] sharing ``seg_of``: one launch; ``sum_out``
    [n_rows, H] also receives the batches' sum."""
    lib = _lib.load()
    if g.dim() != 3 or g.shape != arg.shape or not g.is_contiguous() or not arg.is_contiguous() or g.dtype != torch.float32:
        raise _lib.BganError("scatter_arg_batch: g and arg must be contiguous [n_batch, n_seg, H]")
    if seg_of.dtype != torch.int32 or seg_of.numel() != n_rows:
        raise _lib.BganError("scatter_arg_batch: seg_of must be int32 [n_rows]")
    nb, S, H = g.shape
    out = torch.empty(nb, n_rows, H, dtype=torch.float32, device=g.device)
    if sum_out is not None and (tuple(sum_out.shape) != (n_rows, H) or sum_out.dtype != torch.float32 or not sum_out.is_contiguous()):
        raise _lib.BganError("scatter_arg_batch: sum_out must be a contiguous fp32 [n_rows, H]")
    check(lib.bg_scatter_arg_batch_f32(ptr(g), H, S * H, ptr(arg), S * H, ptr(seg_of), ptr(out), H, n_rows * H, S, n_rows, H,
                                       nb, ptr(sum_out), H, stream()), "bg_scatter_arg_batch_f32")
    return out


def program_ratio_raw(plan, att: torch.Tensor, mask: torch.Tensor, voxel_feature: torch.Tensor, area_col: int):
    """(class ratio [n_slot, 1], program weight [Np, 1], FAR [B, 1]) of util_graph.get_program_ratio without a gradient:
    bg_program_ratio_f32, two launches.  None when the entry point does not take the batch (the caller then runs the
    op-by-op form)."""
    lib = _lib.load()
    ev, ep = plan.edge_voxel_gather.csr, plan.edge_program_gather.T.csr
    sl, sg, gs = plan.slot_sum.csr, plan.slot_graph_sum.csr, plan.slot_graph_gather.csr
    a2, m2 = _f32c(att).reshape(-1), _f32c(mask).reshape(-1)
    if (ev.col is None or ep.col is None or sl.col is None or sg.col is None or gs.col is None or not a2.is_contiguous()
            or not m2.is_contiguous() or a2.numel() != ev.n_rows or m2.numel() != voxel_feature.shape[0]):
        return None
    Np, n_slot, B = ep.n_rows, sl.n_rows, sg.n_rows
    dev = voxel_feature.device
    ratio = torch.empty(n_slot, 1, dtype=torch.float32, device=dev)
    pw = torch.empty(Np, 1, dtype=torch.float32, device=dev)
    far = torch.empty(B, 1, dtype=torch.float32, device=dev)
    rc = lib.bg_program_ratio_f32(ptr(a2), ptr(m2), voxel_feature.data_ptr() + 4 * area_col, voxel_feature.stride(0), ptr(ev.col),
                                  ptr(ep.rowptr), ptr(ep.col), ptr(sl.rowptr), ptr(sl.col), ptr(sg.rowptr), ptr(sg.col),
                                  ptr(gs.col), ptr(ratio), ptr(pw), ptr(far), Np, n_slot, B, stream())
    if rc == _lib.BG_ERR_SHAPE:
        return None
    check(rc, "bg_program_ratio_f32")
    return ratio, pw, far


def gather_arg_raw(x: torch.Tensor, arg: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x2 = _f32c(x)
    S, H = arg.shape
    out = torch.empty(S, H, dtype=torch.float32, device=x.device)
    check(lib.bg_gather_arg_f32(ptr(x2), _ld(x2), ptr(arg), ptr(out), H, S, H, stream()), "bg_gather_arg_f32")
    return out


def scatter_arg_raw(g: torch.Tensor, arg: torch.Tensor, n_rows: int, out: Optional[torch.Tensor] = None,
                    seg_of: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[arg[s, h], h] = g[s, h], zero elsewhere.  ``seg_of`` (int32 [n_rows], the segment of every row; the
    pooling plans carry it as Csr.seg_of): one launch instead of a zero fill + scatter."""
    lib = _lib.load()
    g2 = _f32c(g)
    S, H = arg.shape
    if out is None:
        out = torch.empty(n_rows, H, dtype=torch.float32, device=g.device)
    if seg_of is not None and (seg_of.dtype != torch.int32 or seg_of.numel() != n_rows or not arg.is_contiguous()):
        raise _lib.BganError("scatter_arg: seg_of must be int32 [n_rows] and arg contiguous")
    check(lib.bg_scatter_arg_f32(ptr(g2), _ld(g2), ptr(arg), ptr(seg_of), ptr(out), H, S, n_rows, H, stream()),
          "bg_scatter_arg_f32")
    return out


_CHAIN_LDS_LIMIT = 160 * 1024      # csrc/segment.hip: BG_CHAIN_LDS_LIMIT


def _chain_fits(op: LinOp, blocks, W: int, *ts) -> bool:
    """Can bg_spmm_chain_f32 run this operator?  (square block-diagonal CSR with explicit columns, the
    largest block's rows resident in LDS, 16-byte operands)"""
    c = op.csr
    if blocks is None or c.col is None or c.n_rows != c.n_cols or W % 4 != 0 or W == 0:
        return False
    if blocks.max_rows * 40 + 4 * (blocks.max_rows + 1) > _CHAIN_LDS_LIMIT:
        return False
    return all(t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 for t in ts)


def _chain_launch(op: LinOp, blocks, x0, ldx, add, ld_add, add_step, out, ldo, out_step, n_steps, W, basis=CHAIN_POWER):
    """x0 / add / out are raw device addresses (ints); pitches and steps in floats."""
    lib = _lib.load()
    c = op.csr
    check(lib.bg_spmm_chain_f32(x0, ldx, ptr(c.rowptr), ptr(c.col), ptr(op.row_scale), op.row_mode,
                                ptr(op.col_scale), op.col_mode, ptr(blocks.ptr), blocks.n, blocks.max_rows,
                                blocks.max_nnz, blocks.max_deg, add, ld_add, add_step, out, ldo, out_step, n_steps, W, basis,
                                stream()),
          "bg_spmm_chain_f32")


def power_chain_raw(op: LinOp, Z: torch.Tensor, L: int, W: int, blocks=None, basis=None) -> torch.Tensor:
    """In place on Z [N, (L+1)*W] (or [N, L+1, W]; row pitch = Z.stride(0)): column group k receives
    op^k (basis CHAIN_POWER) or T_k(op) (CHAIN_CHEB; default D_BASIS) applied to column group 0, k = 1..L.  One launch
    when the operator is block diagonal with LDS-sized blocks (plan.vox_blocks), else L launches."""
    _lib.require_device(Z, "power_chain")
    basis = D_BASIS if basis is None else basis
    N = Z.shape[0]
    Z2 = Z.view(N, -1)
    if Z2.dtype != torch.float32 or Z2.stride(1) != 1 or Z2.shape[1] < (L + 1) * W or N != op.csr.n_cols:
        raise _lib.BganError("power_chain: Z must be fp32 [N, >= (L+1)*W] with unit inner stride, N = the plan's rows")
    if L <= 0 or N == 0:
        return Z
    if _chain_fits(op, blocks, W, Z2):
        ld = Z2.stride(0)
        _chain_launch(op, blocks, Z2.data_ptr(), ld, None, 0, 0, Z2.data_ptr() + 4 * W, ld, W, L, W, basis)
        return Z
    for k in range(1, L + 1):
        zk = Z2[:, k * W:(k + 1) * W]
        spmm_raw(op, Z2[:, (k - 1) * W:k * W], out=zk)
        if basis == CHAIN_CHEB and k >= 2:                           # T_k = 2 x T_{k-1} - T_{k-2}
            zk.mul_(2.0).sub_(Z2[:, (k - 2) * W:(k - 1) * W])
    return Z


def power_chain_pair_raw(op: LinOp, Z1: torch.Tensor, W1: int, Z2: torch.Tensor, W2: int, L: int, blocks=None,
                         basis=None) -> None:
    """power_chain_raw on two matrices under the same operator: ONE launch when both fit the short-row chain
    kernel (a chain's time does not depend on its width, so the second one rides along), else two calls."""
    basis = D_BASIS if basis is None else basis
    N = Z1.shape[0]
    A, B = Z1.view(N, -1), Z2.view(N, -1)
    ok = (blocks is not None and L > 0 and N > 0 and blocks.max_deg <= 8 and blocks.max_rows <= (3072 if blocks.max_deg <= 6 else 2048)
          and A.dtype == torch.float32 and B.dtype == torch.float32 and A.stride(1) == 1 and B.stride(1) == 1
          and A.shape[1] >= (L + 1) * W1 and B.shape[1] >= (L + 1) * W2 and B.shape[0] == N and N == op.csr.n_cols
          and _chain_fits(op, blocks, W1, A) and _chain_fits(op, blocks, W2, B))
    if not ok:
        power_chain_raw(op, Z1, L, W1, blocks, basis)
        power_chain_raw(op, Z2, L, W2, blocks, basis)
        return
    _lib.require_device(Z1, "power_chain_pair")
    lib = _lib.load()
    c = op.csr
    check(lib.bg_spmm_chain_pair_f32(A.data_ptr(), A.stride(0), A.data_ptr() + 4 * W1, A.stride(0), W1, W1,
                                     B.data_ptr(), B.stride(0), B.data_ptr() + 4 * W2, B.stride(0), W2, W2,
                                     ptr(c.rowptr), ptr(c.col), ptr(op.row_scale), op.row_mode, ptr(op.col_scale),
                                     op.col_mode, ptr(blocks.ptr), blocks.n, blocks.max_rows, blocks.max_nnz,
                                     blocks.max_deg, L, basis, stream()), "bg_spmm_chain_pair_f32")


def horner_chain_raw(op: LinOp, Y: torch.Tensor, L: int, W: int, blocks=None, basis=None) -> torch.Tensor:
    """sum_k op^k Y_k (CHAIN_POWER) or sum_k T_k(op) Y_k (CHAIN_CHEB; default D_BASIS) for Y [N, (L+1)*W] (column
    group k = Y_k): Horner's rule acc = Y_L; acc = op(acc) + Y_k for k = L-1..0, resp. Clenshaw's
    b_k = Y_k + 2 op(b_{k+1}) - b_{k+2}, result Y_0 + op(b_1) - b_2.  One launch under the conditions of power_chain_raw."""
    _lib.require_device(Y, "horner_chain")
    basis = D_BASIS if basis is None else basis
    N = Y.shape[0]
    Y2 = _f32c(Y.view(N, -1))
    if Y2.shape[1] < (L + 1) * W or N != op.csr.n_cols:
        raise _lib.BganError("horner_chain: Y must be [N, >= (L+1)*W], N = the plan's rows")
    if L <= 0 or N == 0:
        return Y2[:, L * W:(L + 1) * W].clone()
    out = torch.empty(N, W, dtype=torch.float32, device=Y.device)
    if _chain_fits(op, blocks, W, Y2, out):
        ld = Y2.stride(0)
        base = Y2.data_ptr()
        _chain_launch(op, blocks, base + 4 * L * W, ld, base + 4 * (L - 1) * W, ld, -W, out.data_ptr(), W, 0, L, W,
                      CHAIN_CLENSHAW if basis == CHAIN_CHEB else CHAIN_POWER)
        return out
    acc = Y2[:, L * W:(L + 1) * W]
    if basis != CHAIN_CHEB:
        for k in range(L - 1, -1, -1):
            acc = spmm_add_raw(op, acc, Y2[:, k * W:(k + 1) * W], out=out if k == 0 else None)
        return out
    b1, b2 = acc, None                                               # b_{k+1}, b_{k+2}
    for k in range(L - 1, -1, -1):
        t = spmm_raw(op, b1)
        if k > 0:
            t.mul_(2.0)
        if b2 is not None:
            t.sub_(b2)
        t.add_(Y2[:, k * W:(k + 1) * W])
        b1, b2 = t, b1
    out.copy_(b1)
    return out


_PtrArr4 = C.c_void_p * 4
_I64Arr4 = C.c_int64 * 4


_GEMM_TLS = threading.local()


def _gemm_arrays():
    """Per-thread ctypes argument arrays of bg_gemm_f32 (the autograd engine calls backward functions from its
    own thread); building three arrays per call cost more than the launch itself."""
    t = _GEMM_TLS
    if not hasattr(t, "pa"):
        t.pa, t.la, t.ka = _PtrArr4(), _I64Arr4(), _I64Arr4()
    return t.pa, t.la, t.ka


def gemm_nbr_raw(A: Sequence[torch.Tensor], W: torch.Tensor, pairs: torch.Tensor, nbr, rows: torch.Tensor,
                 act: int = ACT_NONE, slope: float = 0.01, residual: Optional[torch.Tensor] = None,
                 acc_init: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 pre_residual_out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """bg_gemm_nbr_f32: C = epilogue(acc_init + cat(A, -1) W^T + addend[rows]) with the addend formed on load from
    ``pairs`` [G, 2N] = [own part | neighbour part] and the LinOp ``nbr`` over its rows (mean over the neighbour rows +
    the own part: what spmm_add_raw(nbr, pairs[:, N:], pairs[:, :N]) would write).  Returns None when the library
    declines the shape (only the row-panel kernel does this): the caller then issues the two launches."""
    lib = _lib.load()
    N = W.shape[0]
    keep = [_f32c(a) for a in A]
    W2 = _f32c(W)
    if pairs.dtype != torch.float32 or pairs.stride(1) != 1 or pairs.shape[1] != 2 * N or nbr.row_mode != SCALE_DIV:
        return None
    pa, la, ka = _gemm_arrays()
    for i in range(4):
        if i < len(keep):
            pa[i], la[i], ka[i] = keep[i].data_ptr(), keep[i].stride(0), keep[i].shape[1]
        else:
            pa[i], la[i], ka[i] = None, 0, 0
    M = keep[0].shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=W.device)
    ptr = _lib.ptr
    rc = lib.bg_gemm_nbr_f32(M, N, len(keep), pa, la, ka, ptr(W2), W2.stride(0), ptr(out), out.stride(0), None, act, slope,
                             ptr(residual), residual.stride(0) if residual is not None else 0,
                             ptr(pairs), pairs.stride(0), pairs.data_ptr() + 4 * N, pairs.stride(0),
                             ptr(nbr.csr.rowptr), ptr(nbr.csr.col), ptr(nbr.row_scale), ptr(rows),
                             ptr(acc_init), acc_init.stride(0) if acc_init is not None else 0,
                             ptr(pre_residual_out), pre_residual_out.stride(0) if pre_residual_out is not None else 0, stream())
    if rc == _lib.BG_ERR_SHAPE:
        return None
    check(rc, "bg_gemm_nbr_f32")
    return out


def gemm_raw(A: Sequence[torch.Tensor], B: torch.Tensor, transA: bool, transB: bool,
             bias: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
             act: int = ACT_NONE, slope: float = 0.01, residual: Optional[torch.Tensor] = None,
             pre_add: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
             pre_rows: Optional[torch.Tensor] = None, acc_init: Optional[torch.Tensor] = None,
             pre_residual_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(acc_init + cat(A, -1 (or 0 when transA)) (x) B); see bg_gemm_f32."""
    lib = _lib.load()
    n_seg = len(A)
    if n_seg < 1 or n_seg > 4:
        raise _lib.BganError("gemm: 1..4 A segments")
    B2 = _f32c(B)
    _lib.require_device(B2, "gemm")
    pa, la, ka = _gemm_arrays()
    M = -1
    Ktot = 0
    keep = []                                   # converted operands must outlive the launch call
    for i in range(4):
        if i < n_seg:
            a = _f32c(A[i])
            keep.append(a)
            r, c = a.shape
            m, k = (c, r) if transA else (r, c)
            if M < 0:
                M = m
            elif m != M:
                raise _lib.BganError("gemm: A segments disagree on M")
            pa[i], la[i], ka[i] = a.data_ptr(), (a.stride(0) if r > 1 else max(c, 1)), k
            Ktot += k
        else:
            pa[i], la[i], ka[i] = None, 0, 0
    N = B2.shape[0] if transB else B2.shape[1]
    if (B2.shape[1] if transB else B2.shape[0]) != Ktot:
        raise _lib.BganError(f"gemm: reduction sizes differ ({Ktot} vs {tuple(B2.shape)}, transB={transB})")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=B2.device)
    elif tuple(out.shape) != (M, N) or out.dtype != torch.float32 or out.stride(-1) != 1:
        raise _lib.BganError("gemm: `out` must be an fp32 [M, N] matrix with unit inner stride")
    if M == 0 or N == 0:
        return out
    if bias is not None:
        bias = _f32c(bias)
    if row_scale is not None:
        row_scale = _f32c(row_scale.reshape(-1))
    ldr = ldp = 0
    if residual is not None:
        residual = _f32c(residual)
        ldr = _ld(residual)
    if pre_add is not None:
        pre_add = _f32c(pre_add)
        ldp = _ld(pre_add)
        if pre_rows is not None and (pre_rows.dtype != torch.int32 or pre_rows.numel() != M):
            raise _lib.BganError("gemm: pre_rows must hold one int32 row index per output row")
    ld0 = 0
    if acc_init is not None:
        acc_init = _f32c(acc_init)
        if tuple(acc_init.shape) != (M, N):
            raise _lib.BganError("gemm: acc_init must be an [M, N] matrix")
        ld0 = _ld(acc_init)
    if pre_residual_out is not None and (tuple(pre_residual_out.shape) != (M, N) or pre_residual_out.stride(-1) != 1
                                         or pre_residual_out.dtype != torch.float32):
        raise _lib.BganError("gemm: pre_residual_out must be an fp32 [M, N] matrix with unit inner stride")
    check(lib.bg_gemm_f32(int(transA), int(transB), M, N, n_seg, pa, la, ka, ptr(B2), _ld(B2), ptr(out), _ld(out),
                          ptr(bias), ptr(row_scale), act, slope, ptr(residual), ldr, ptr(pre_add), ldp,
                          ptr(pre_rows), ptr(acc_init), ld0, ptr(pre_residual_out),
                          _ld(pre_residual_out) if pre_residual_out is not None else 0, stream()), "bg_gemm_f32")
    return out


def gemm_tn_raw(A: torch.Tensor, B, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """C[M,N] = A^T [B_0|B_1|..] for A [R,M], B_s [R,N_s] with a long R: split-K, deterministic order.
    ``out`` + ``accumulate``: add the product to an existing [M,N] buffer."""
    lib = _lib.load()
    A2 = _f32c(A)
    Bs = [_f32c(b) for b in ([B] if torch.is_tensor(B) else list(B))]
    R, M = A2.shape
    if len(Bs) < 1 or len(Bs) > 4 or any(b.shape[0] != R for b in Bs):
        raise _lib.BganError("gemm_tn: 1..4 B segments with the same reduction length as A")
    Ns = [b.shape[1] for b in Bs]
    N = sum(Ns)
    if out is None:
        if accumulate:
            raise _lib.BganError("gemm_tn: accumulate needs the buffer to add to")
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    elif tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.stride(0) < N or out.dtype != torch.float32:
        raise _lib.BganError("gemm_tn: `out` must be an fp32 [M, N] matrix with contiguous rows")
    if M == 0 or N == 0:
        return out
    tiles = ((M + 63) // 64) * sum((n + 63) // 64 for n in Ns)
    steps = (R + 31) // 32
    target = 4096 if R >= 65536 else (1024 if R >= 32768 else 512)       # csrc/driver.hip tn_target: long reductions split finer
    splits = max(1, min(steps, (target + tiles - 1) // tiles, 256))
    partial = torch.empty(splits, M, N, dtype=torch.float32, device=A.device)
    pb = _PtrArr4(*([b.data_ptr() for b in Bs] + [None] * (4 - len(Bs))))
    lb = _I64Arr4(*([_ld(b) for b in Bs] + [0] * (4 - len(Bs))))
    nb = _I64Arr4(*(Ns + [0] * (4 - len(Bs))))
    check(lib.bg_gemm_tn_splitk_f32(M, R, ptr(A2), _ld(A2), len(Bs), pb, lb, nb, ptr(out), out.stride(0), ptr(partial),
                                    splits, int(accumulate), stream()), "bg_gemm_tn_splitk_f32")
    return out


def colsum_raw(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    lib = _lib.load()
    x2 = _f32c(x)
    M, N = x2.shape
    nb = max(1, min(256, (M + 127) // 128))
    partial = torch.empty(nb, N, dtype=torch.float32, device=x.device)
    if out is None:
        if accumulate:
            raise _lib.BganError("colsum: accumulate needs the buffer to add to")
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    elif out.numel() != N or not out.is_contiguous() or out.dtype != torch.float32:
        raise _lib.BganError("colsum: `out` must be a contiguous fp32 vector of N elements")
    check(lib.bg_colsum_f32(ptr(x2), _ld(x2), M, N, ptr(partial), nb, ptr(out), int(accumulate), stream()), "bg_colsum_f32")
    return out


# ------------------------------------------------------------------------------- autograd family
class ManualCtx:
    """What an autograd Function's forward / backward use of their ``ctx``, for calling them directly (no graph, no
    engine pass): models.DiscriminatorVoxel.critic_objective(manual=True)."""

    def __init__(self, n_inputs: int):
        self.needs_input_grad = (True,) * n_inputs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


class SpMM(torch.autograd.Function):
    """y = op(x) for a fixed sparse LinOp; backward is the transposed LinOp."""

    @staticmethod
    def forward(ctx, x, op: LinOp):
        ctx.op = op
        return spmm_raw(op, x)

    @staticmethod
    def backward(ctx, g):
        return SpMM.apply(g, ctx.op.T), None


class SpMMAdd(torch.autograd.Function):
    """y = op(x) + a in one pass (short-row plans)."""

    @staticmethod
    def forward(ctx, x, a, op: LinOp):
        ctx.op = op
        return spmm_add_raw(op, x, a)

    @staticmethod
    def backward(ctx, g):
        gx = SpMM.apply(g, ctx.op.T) if ctx.needs_input_grad[0] else None
        return gx, (g if ctx.needs_input_grad[1] else None), None


class PowerChain(torch.autograd.Function):
    """Z [N, L+1, W] = (x, op x, .., op^L x), or (T_0(op) x, .., T_L(op) x) in the Chebyshev basis; the backward is a
    HornerChain over the transposed operator."""

    @staticmethod
    def forward(ctx, x, op: LinOp, L: int, blocks, basis=None):
        N, W = x.shape
        basis = D_BASIS if basis is None else basis
        Z = torch.empty(N, L + 1, W, dtype=torch.float32, device=x.device)
        Z[:, 0].copy_(x)
        power_chain_raw(op, Z, L, W, blocks, basis)
        ctx.op, ctx.L, ctx.blocks, ctx.basis = op, L, blocks, basis
        return Z

    @staticmethod
    def backward(ctx, G):
        return HornerChain.apply(G, ctx.op.T, ctx.blocks, ctx.basis), None, None, None, None


class HornerChain(torch.autograd.Function):
    """sum_k op^k Y[:, k] (or sum_k T_k(op) Y[:, k]) for Y [N, L+1, W]; the backward is a PowerChain over the
    transposed operator."""

    @staticmethod
    def forward(ctx, Y, op: LinOp, blocks, basis=None):
        N, L1, W = Y.shape
        basis = D_BASIS if basis is None else basis
        ctx.op, ctx.L, ctx.blocks, ctx.basis = op, L1 - 1, blocks, basis
        return horner_chain_raw(op, Y.contiguous(), L1 - 1, W, blocks, basis)

    @staticmethod
    def backward(ctx, g):
        return PowerChain.apply(g.contiguous(), ctx.op.T, ctx.L, ctx.blocks, ctx.basis), None, None, None


class SpMMW(torch.autograd.Function):
    """y[i] = sum_e w[e] x[col(e)] with differentiable per-edge weights (original edge order)."""

    @staticmethod
    def forward(ctx, x, w, op: LinOp):
        ctx.op = op
        ctx.save_for_backward(x, w)
        return spmm_raw(op, x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = SpMMW.apply(g, w, ctx.op.T) if ctx.needs_input_grad[0] else None
        gw = SDDMM.apply(g, x, ctx.op).view_as(w) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class SDDMM(torch.autograd.Function):
    """out[e] = <a[row(e)], b[col(e)]>."""

    @staticmethod
    def forward(ctx, a, b, op: LinOp):
        ctx.op = op
        ctx.save_for_backward(a, b)
        return sddmm_raw(op, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = SpMMW.apply(b, g, ctx.op) if ctx.needs_input_grad[0] else None
        gb = SpMMW.apply(a, g, ctx.op.T) if ctx.needs_input_grad[1] else None
        return ga, gb, None


class MatMul(torch.autograd.Function):
    """C = op(A) op(B) on the fp32 matrix cores; closed under differentiation."""

    @staticmethod
    def forward(ctx, A, B, ta: bool, tb: bool):
        ctx.ta, ctx.tb = ta, tb
        ctx.save_for_backward(A, B)
        if ta and not tb:
            return gemm_tn_raw(A, B)
        return gemm_raw([A], B, ta, tb)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        gA = gB = None
        if ctx.needs_input_grad[0]:
            if not ta:
                gA = MatMul.apply(g, B, False, not tb)      # g B^T (tb=False) / g B (tb=True)
            else:
                gA = MatMul.apply(B, g, tb, True)           # B g^T / B^T g^T
        if ctx.needs_input_grad[1]:
            if not tb:
                gB = MatMul.apply(A, g, not ta, False)      # A^T g / A g
            else:
                gB = MatMul.apply(g, A, True, ta)           # g^T A / g^T A^T
        return gA, gB, None, None


class CatTN(torch.autograd.Function):
    """dW = g^T [x_0 | x_1 | ..]  (the weight gradient of LinearCat) in ONE split-K launch."""

    @staticmethod
    def forward(ctx, g, *xs):
        ctx.widths = [x.shape[1] for x in xs]
        ctx.save_for_backward(g, *xs)
        return gemm_tn_raw(g, list(xs))

    @staticmethod
    def backward(ctx, G2):
        g, *xs = ctx.saved_tensors
        gg = LinearCat.apply(G2, None, ACT_NONE, 0.0, None, None, None, *xs) if ctx.needs_input_grad[0] else None
        gxs: List[Optional[torch.Tensor]] = [None] * len(xs)
        if any(ctx.needs_input_grad[1:]):
            gx = MatMul.apply(g, G2, False, False)
            off = 0
            for i, wdt in enumerate(ctx.widths):
                if ctx.needs_input_grad[1 + i]:
                    gxs[i] = gx[:, off:off + wdt]
                off += wdt
        return (gg, *gxs)


class ColSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.rows = x.shape[0]
        return colsum_raw(x)

    @staticmethod
    def backward(ctx, g):
        return g.unsqueeze(0).expand(ctx.rows, -1)


class LinearCat(torch.autograd.Function):
    """y = act(row_scale * (cat(xs, -1) @ W^T + b) + pre_add) + residual: the fused nn.Linear of the
    hot path (every epilogue term optional; pre_add and residual are differentiable inputs)."""

    @staticmethod
    def forward(ctx, W, b, act, slope, row_scale, residual, pre_add, *xs):
        # the residual can ride in the epilogue unless a backward will need the activation's own sign
        global _NO_GRAD_CALL
        no_backward = _NO_GRAD_CALL or not any(ctx.needs_input_grad)
        _NO_GRAD_CALL = False                   # one-shot: a direct LinearCat.apply never inherits it
        fuse_res = residual if (act != ACT_LEAKY or no_backward) else None
        w_t = isinstance(act, tuple)            # (act, "t"): W is stored [K, out] instead of [out, K]
        if w_t:
            act = act[0]
        ctx.w_t = w_t
        pre_op = None
        if isinstance(slope, tuple):            # (slope, gather LinOp): pre_add has one row per gathered group
            slope, pre_op = slope
        ctx.pre_op = pre_op
        y = gemm_raw(list(xs), W, False, not w_t, bias=b, row_scale=row_scale, act=act, slope=slope,
                     residual=fuse_res, pre_add=pre_add, pre_rows=pre_op.csr.col if pre_op is not None else None)
        ctx.act, ctx.slope = act, slope
        ctx.inputs_only = _INPUT_GRADS_ONLY
        ctx.has_res, ctx.has_pre = residual is not None, pre_add is not None
        ctx.widths = [x.shape[1] for x in xs]
        # leaky(z) has the sign of z: the activation's output doubles as the mask for the backward
        act_out = y if act == ACT_LEAKY else None
        if residual is not None and fuse_res is None:
            y = y + residual
        ctx.save_for_backward(W, row_scale, act_out, *xs)
        return y

    @staticmethod
    def backward(ctx, g):
        W, row_scale, act_out, *xs = ctx.saved_tensors
        g_res = g if ctx.has_res and ctx.needs_input_grad[5] else None
        if ctx.act == ACT_LEAKY:
            if torch.is_grad_enabled():
                g = torch.where(act_out > 0, g, g * ctx.slope)                        # differentiable form
            else:
                g = torch.ops.aten.leaky_relu_backward(g, act_out, ctx.slope, True)   # one launch; input = the result
        g_pre = g if ctx.has_pre and ctx.needs_input_grad[6] else None
        if g_pre is not None and ctx.pre_op is not None:
            g_pre = SpMM.apply(g_pre, ctx.pre_op.T)          # transpose of the fused gather: sum over each group
        if row_scale is not None:
            g = g * row_scale.view(-1, 1)
        gW = gb = None
        if ctx.needs_input_grad[0] and not ctx.inputs_only:
            gW = CatTN.apply(g, *xs)
            if ctx.w_t:
                gW = gW.t()
        if ctx.needs_input_grad[1] and not ctx.inputs_only:
            gb = ColSum.apply(g)
        gxs: List[Optional[torch.Tensor]] = [None] * len(xs)
        need = [i for i in range(len(xs)) if ctx.needs_input_grad[7 + i]]
        if need:
            # only the column range that covers the segments wanting a gradient (data segments such as
            # the position terms sit at the end: one column tile less)
            offs = [0]
            for wdt in ctx.widths:
                offs.append(offs[-1] + wdt)
            lo, hi = offs[need[0]], offs[need[-1] + 1]
            Wsub = W if (lo == 0 and hi == offs[-1]) else (W[lo:hi] if ctx.w_t else W[:, lo:hi])
            gx = MatMul.apply(g, Wsub, False, ctx.w_t)
            if len(need) == 1:
                gxs[need[0]] = gx
            else:
                parts = gx.split([offs[i + 1] - offs[i] for i in range(need[0], need[-1] + 1)], dim=1)
                for i in need:
                    gxs[i] = parts[i - need[0]]
        return (gW, gb, None, None, None, g_res, g_pre, *gxs)


def linear(xs, W, b=None, act=ACT_NONE, slope=0.01, row_scale=None, residual=None, pre_add=None, w_t=False,
           pre_gather: Optional[LinOp] = None):
    """``w_t``: W is given as [K, out] (y = cat(xs) @ W) instead of the nn.Linear layout [out, K].
    ``pre_gather``: a pure gather LinOp (one source row per output row, no scaling); ``pre_add`` then holds
    one row per SOURCE and the gather is fused into the product's epilogue."""
    global _NO_GRAD_CALL
    _NO_GRAD_CALL = not torch.is_grad_enabled()       # Function.forward itself always runs with grad mode off
    if torch.is_tensor(xs):
        xs = [xs]
    if pre_gather is not None:
        c = pre_gather.csr
        if pre_gather.row_mode != SCALE_NONE or pre_gather.col_mode != SCALE_NONE or c.nnz != c.n_rows:
            raise _lib.BganError("linear: pre_gather must be an unscaled one-entry-per-row gather")
    return LinearCat.apply(W, b, (act, "t") if w_t else act, (slope, pre_gather) if pre_gather is not None else slope,
                           row_scale, residual, pre_add, *xs)


class SegMax(torch.autograd.Function):
    """Segment max over a pooling Csr; the gradient goes to the (first) arg-max row only."""

    @staticmethod
    def forward(ctx, x, csr: Csr):
        out, arg = segmax_raw(csr, x)
        ctx.n_rows = x.shape[0]
        ctx.seg_of = csr.seg_of
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)       # no zero int tensor for the arg output's absent gradient
        return out, arg

    @staticmethod
    def backward(ctx, g, _garg):
        (arg,) = ctx.saved_tensors
        if g is None:
            return None, None
        return ScatterArg.apply(g, arg, ctx.n_rows, ctx.seg_of), None


class ScatterArg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, arg, n_rows, seg_of=None):
        ctx.save_for_backward(arg)
        ctx.seg_of = seg_of
        return scatter_arg_raw(g, arg, n_rows, seg_of=seg_of)

    @staticmethod
    def backward(ctx, gg):
        (arg,) = ctx.saved_tensors
        return GatherArg.apply(gg, arg, ctx.seg_of), None, None, None


class ScatterArgPair(torch.autograd.Function):
    """[scatter(g1, arg1); scatter(g2, arg2)] stacked into one [2 * n_rows, H] matrix (no concatenation)."""

    @staticmethod
    def forward(ctx, g1, arg1, g2, arg2, n_rows, seg_of1=None, seg_of2=None):
        ctx.n_rows = n_rows
        ctx.seg = (seg_of1, seg_of2)
        ctx.save_for_backward(arg1, arg2)
        out = torch.empty(2 * n_rows, arg1.shape[1], dtype=torch.float32, device=g1.device)
        scatter_arg_raw(g1, arg1, n_rows, out=out[:n_rows], seg_of=seg_of1)
        scatter_arg_raw(g2, arg2, n_rows, out=out[n_rows:], seg_of=seg_of2)
        return out

    @staticmethod
    def backward(ctx, gg):
        arg1, arg2 = ctx.saved_tensors
        n = ctx.n_rows
        return (GatherArg.apply(gg[:n], arg1, ctx.seg[0]), None, GatherArg.apply(gg[n:], arg2, ctx.seg[1]), None, None,
                None, None)


class GatherArg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, arg, seg_of=None):
        ctx.n_rows = x.shape[0]
        ctx.seg_of = seg_of
        ctx.save_for_backward(arg)
        return gather_arg_raw(x, arg)

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        return ScatterArg.apply(g, arg, ctx.n_rows, ctx.seg_of), None, None


# ------------------------------------------------------------------------------- attention ops
class CrossLogits(torch.autograd.Function):
    """att[e] = theta . tanh(a[cev[e]] + b[cep[e]])  (first-order autograd only)."""

    @staticmethod
    def forward(ctx, a, b, theta, plan):
        lib = _lib.load()
        cv = plan.cross_v
        a2, b2, th = _f32c(a), _f32c(b), _f32c(theta.reshape(-1))
        att = torch.empty(cv.nnz, dtype=torch.float32, device=a.device)
        check(lib.bg_cross_logits_f32(ptr(a2), _ld(a2), ptr(b2), _ld(b2), ptr(th), ptr(cv.rowptr), ptr(cv.col),
                                      ptr(att), cv.n_rows, a2.shape[1], stream()), "bg_cross_logits_f32")
        ctx.plan = plan
        ctx.save_for_backward(a2, b2, th)
        ctx.theta_shape = theta.shape
        return att

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        a2, b2, th = ctx.saved_tensors
        plan = ctx.plan
        cv = plan.cross_v
        H = a2.shape[1]
        dev = a2.device
        g = _f32c(g.reshape(-1))
        nb = max(1, min(1024, (cv.n_rows + 3) // 4))
        dpre = torch.empty(cv.nnz, H, dtype=torch.float32, device=dev)
        da = torch.empty(cv.n_rows, H, dtype=torch.float32, device=dev)
        dth_p = torch.empty(nb, H, dtype=torch.float32, device=dev)
        check(lib.bg_cross_logits_bwd_f32(ptr(a2), _ld(a2), ptr(b2), _ld(b2), ptr(th), ptr(cv.rowptr), ptr(cv.col),
                                          ptr(g), ptr(dpre), ptr(da), H, ptr(dth_p), nb, cv.n_rows, H, stream()),
              "bg_cross_logits_bwd_f32")
        db = spmm_raw(plan.edge_program_gather.T, dpre)
        dth = colsum_raw(dth_p).view(ctx.theta_shape)
        return da, db, dth, None


class SegGumbelSoftmax(torch.autograd.Function):
    """(soft, hard, arg) of the segment Gumbel-softmax pointer; hard carries the straight-through
    gradient of soft (reference util.py:26)."""

    @staticmethod
    def forward(ctx, att, expn, plan, tau):
        lib = _lib.load()
        cv = plan.cross_v
        att2, e2 = _f32c(att.reshape(-1)), _f32c(expn.reshape(-1))
        if att2.numel() != cv.nnz or e2.numel() != cv.nnz:
            raise _lib.BganError("segment gumbel softmax: logits / noise length differ from the plan")
        soft = torch.empty_like(att2)
        hard = torch.empty_like(att2)
        arg = torch.empty(cv.n_rows, dtype=torch.int32, device=att.device)
        check(lib.bg_seg_gumbel_softmax_f32(ptr(att2), ptr(e2), ptr(cv.rowptr), float(tau), ptr(soft), ptr(hard),
                                            ptr(arg), cv.n_rows, stream()), "bg_seg_gumbel_softmax_f32")
        ctx.plan, ctx.tau = plan, float(tau)
        ctx.save_for_backward(soft)
        ctx.mark_non_differentiable(arg)
        return soft, hard, arg

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_soft, g_hard, _g_arg):
        lib = _lib.load()
        (soft,) = ctx.saved_tensors
        cv = ctx.plan.cross_v
        g = _f32c((g_soft + g_hard).reshape(-1))
        d_att = torch.empty_like(soft)
        check(lib.bg_seg_softmax_bwd_f32(ptr(soft), ptr(g), ptr(cv.rowptr), ctx.tau, ptr(d_att), cv.n_rows, stream()),
              "bg_seg_softmax_bwd_f32")
        return d_att, None, None, None


class MaskGumbel(torch.autograd.Function):
    """(soft[:, 0], hard[:, 0]) of the two-way Gumbel-softmax over the mask decoder's logits
    (reference models.py:109-111) in one launch; first-order autograd."""

    @staticmethod
    def forward(ctx, logits, expn):
        lib = _lib.load()
        _lib.require_device(logits, "mask_gumbel")
        l2, e2 = _f32c(logits), _f32c(expn).contiguous()          # logits may be a column slice (row pitch)
        n = l2.shape[0]
        if l2.shape[1] != 2 or tuple(e2.shape) != tuple(l2.shape):
            raise _lib.BganError("mask_gumbel: logits and noise must both be [n, 2]")
        soft0 = torch.empty(n, 1, dtype=torch.float32, device=logits.device)
        hard0 = torch.empty(n, 1, dtype=torch.float32, device=logits.device)
        check(lib.bg_mask_gumbel_f32(ptr(l2), _ld(l2), ptr(e2), ptr(soft0), ptr(hard0), n, stream()),
              "bg_mask_gumbel_f32")
        ctx.save_for_backward(soft0)
        return soft0, hard0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_soft, g_hard):
        (soft0,) = ctx.saved_tensors
        lib = _lib.load()
        n = soft0.shape[0]
        gs = _f32c(g_soft).contiguous() if g_soft is not None else None
        gh = _f32c(g_hard).contiguous() if g_hard is not None else None
        d = torch.empty(n, 2, dtype=torch.float32, device=soft0.device)
        check(lib.bg_mask_gumbel_bwd_f32(ptr(soft0), ptr(gs), ptr(gh), None, ptr(d), 2, n, stream()), "bg_mask_gumbel_bwd_f32")
        return d, None


def story_project_raw(c: Csr, x: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """[mean_s(x) W[:, :H]^T | mean_s(x) W[:, H:]^T] per segment of ``c`` (see bg_story_project_f32)."""
    lib = _lib.load()
    x2, W2 = _f32c(x), _f32c(W)
    H, Hout = x2.shape[1], W2.shape[0]
    if W2.shape[1] != 2 * H or c.count is None or x2.shape[0] != c.n_cols:
        raise _lib.BganError("story_project: W must be [Hout, 2H] and the plan must carry segment counts")
    out = torch.empty(c.n_rows, 2 * Hout, dtype=torch.float32, device=x.device)
    check(lib.bg_story_project_f32(ptr(x2), _ld(x2), ptr(c.rowptr), ptr(c.col), ptr(c.count), ptr(W2), _ld(W2), Hout,
                                   ptr(out), _ld(out), c.n_rows, H, stream()), "bg_story_project_f32")
    return out


def gather_story_raw(op: LinOp, c: Csr, x: torch.Tensor, W: torch.Tensor):
    """(op(x), story_project_raw(c, x, W)) in one launch; ``op`` must be an unweighted row-scaled operator with
    explicit columns (the voxel gather-mean)."""
    lib = _lib.load()
    _lib.require_device(x, "gather_story")
    x2, W2 = _f32c(x), _f32c(W)
    H, Hout = x2.shape[1], W2.shape[0]
    oc = op.csr
    if (oc.col is None or op.col_mode != SCALE_NONE or x2.shape[0] != oc.n_cols or x2.shape[0] != c.n_cols
            or W2.shape[1] != 2 * H or c.count is None):
        raise _lib.BganError("gather_story: operator / segments / weight do not fit the operand")
    gm = torch.empty(oc.n_rows, H, dtype=torch.float32, device=x.device)
    pq = torch.empty(c.n_rows, 2 * Hout, dtype=torch.float32, device=x.device)
    check(lib.bg_gather_story_f32(ptr(x2), _ld(x2), ptr(oc.rowptr), ptr(oc.col), ptr(op.row_scale), op.row_mode,
                                  ptr(gm), H, oc.n_rows, ptr(c.rowptr), ptr(c.col), ptr(c.count), ptr(W2), _ld(W2), Hout,
                                  ptr(pq), 2 * Hout, c.n_rows, H, stream()), "bg_gather_story_f32")
    return gm, pq


def story_bwd_raw(c: Csr, g: torch.Tensor, y: torch.Tensor, v: torch.Tensor, slope: float):
    """(gz, gterm, story) of bg_story_bwd_f32: leaky-ReLU backward from its result, its per-story sums, and the
    per-story means of v, in one launch.  ``c``: story segments that partition the voxel rows."""
    lib = _lib.load()
    _lib.require_device(g, "story_bwd")
    g2, y2, v2 = _f32c(g), _f32c(y), _f32c(v)
    N, H = g2.shape
    if (c.count is None or c.n_cols != N or c.nnz != N or tuple(y2.shape) != (N, H) or tuple(v2.shape) != (N, H)):
        raise _lib.BganError("story_bwd: the story segments must partition the rows of g, y and v")
    gz = torch.empty(N, H, dtype=torch.float32, device=g.device)
    gterm = torch.empty(c.n_rows, H, dtype=torch.float32, device=g.device)
    story = torch.empty(c.n_rows, H, dtype=torch.float32, device=g.device)
    check(lib.bg_story_bwd_f32(ptr(g2), _ld(g2), ptr(y2), _ld(y2), ptr(v2), _ld(v2), ptr(c.rowptr), ptr(c.col),
                               ptr(c.count), float(slope), ptr(gz), H, ptr(gterm), H, ptr(story), H, c.n_rows, H,
                               stream()), "bg_story_bwd_f32")
    return gz, gterm, story


def _identity_op(plan) -> LinOp:
    """Row i <- row i (one slot per row, no column list): lets bg_sddmm_f32 take per-row dot products."""
    op = plan.ops.get("vox_identity")
    if op is None:
        n = plan.Nv
        op = LinOp(Csr(iota_i32(n + 1, plan.device), None, None, n, n, n))
        plan.ops["vox_identity"] = op
    return op


class SharedGrads:
    """Gradient buffers of the weights that the nodes of ONE forward share (the generator applies the same block
    and the same attention module 12 and 6 times, reference models.py:196-201).  Every node adds its share to
    the buffer inside its own weight-gradient launch (``accumulate``) and only the node of the FIRST call --
    the last to run in backward, since the calls form a chain through the voxel features -- hands the sum to
    autograd; without it autograd sums the per-call gradients in one add launch each."""

    def __init__(self):
        self.buf = {}
        self._armed = False

    def enter(self):
        """Backward: a node that shares these buffers runs.  Only the first call's node hands the sums to autograd
        (the calls chain through the voxel features, which makes it the last to run), so whatever a backward pass
        accumulates must have been handed over when the pass ends -- otherwise weight gradients would be dropped
        without a word (the chain was cut by a detach, or a backward restricted by ``inputs=`` stopped short of the
        first call).  A call whose output is not used (the generator's last block call, reference
        models.py:196-203) simply does not run and contributes nothing, correctly."""
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self):
        left = sorted(self.buf)
        self.buf.clear()
        self._armed = False
        if left:
            raise _lib.BganError("shared weight gradients: this backward pass accumulated " + ", ".join(left) +
                                 " but never reached the first call's node, which hands them to autograd; the "
                                 "gradients would have been dropped (set models.SHARE_WEIGHT_GRADS = False for "
                                 "partial backward passes)")

    def slot(self, key, like: torch.Tensor):
        """(buffer, accumulate): the first request of a backward pass creates the buffer (to be overwritten)."""
        t = self.buf.get(key)
        if t is not None:
            return t, True
        t = torch.empty(like.shape, dtype=torch.float32, device=like.device)
        self.buf[key] = t
        return t, False

    def take(self, key, first: bool):
        return self.buf.pop(key, None) if first else None


def _wgrad(shared, key, first, like, fn):
    """fn(out, accumulate) computes one weight gradient; with ``shared`` it lands in the common buffer."""
    if shared is None:
        return fn(None, False)
    buf, acc = shared.slot(key, like)
    fn(buf.view(-1) if buf.dim() != 2 else buf, acc)
    return shared.take(key, first)


FUSE_ATTENTION_BWD_HEAD = True      # AttentionNode.backward: everything up to the softmax / mask backward in one launch


class AttentionNode(torch.autograd.Function):
    """One Attention call (reference models.py:104-126) as ONE autograd node: the forward is the fused pair of
    the graph-free pass (NN_v product, then one launch for mask, cross logits, segment Gumbel-softmax pointer,
    weighted gather and residual), the backward is written by hand over the existing backward kernels --
    17-19 launches for the 29 of the op-by-op graph, 2 for 6 forward.  First-order autograd.

    Inputs: v [Nv,H], x [Np,H] program features, b = NN_p(x) [Np,H], Wcat [H+4,H] / bcat [H+4] (NN_v with the
    composed two-row mask decoder stacked under it), theta, Exp(1) noise.  Outputs (soft0 [Nv,1], hard0 [Nv,1],
    soft [Ec], hard [Ec], new voxel features)."""

    @staticmethod
    def forward(ctx, v, x, b, Wcat, bcat, theta, exp_mask, exp_att, plan, tau: float, shared=None, first=True):
        H = v.shape[1]
        ctx.shared, ctx.first = shared, first
        a = gemm_raw([v], Wcat[:H], False, True, bias=bcat[:H])
        soft0, hard0, soft, hard, out = attention_forward_raw(a, None, b, x, v, theta, plan, exp_mask, exp_att, tau,
                                                              wd=Wcat[H:H + 2], bd=bcat[H:H + 2])
        ctx.plan, ctx.tau = plan, float(tau)
        ctx.theta_shape = theta.shape
        ctx.save_for_backward(v, x, b, a, Wcat, theta, soft0, soft)
        ctx.set_materialize_grads(False)
        return soft0, hard0, soft, hard, out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_soft0, g_hard0, g_soft, g_hard, g_out):
        lib = _lib.load()
        v, x, b, a, Wcat, theta, soft0, soft = ctx.saved_tensors
        if ctx.shared is not None:
            ctx.shared.enter()
        plan = ctx.plan
        cv = plan.cross_v
        N, H = v.shape
        dev = v.device
        # ---- new_v = v + soft0 * gathered,  gathered = sum_e soft_e x[cep_e]
        g_e = None
        g_x = None
        g_more = None
        if (FUSE_ATTENTION_BWD_HEAD and g_out is not None and H % 64 == 0 and H <= 512 and 0 < cv.max_len <= 64
                and cv.col is not None):
            # everything up to the softmax's and the mask's backward in ONE launch (one wave per voxel)
            g_out = _f32c(g_out)
            x2 = _f32c(x)
            G_al = torch.empty(N, H + 4, dtype=torch.float32, device=dev)
            gg = torch.empty(N, H, dtype=torch.float32, device=dev)
            d_att = torch.empty_like(soft)
            cont = lambda t: _f32c(t).reshape(-1).contiguous() if t is not None else None
            gse, ghe, gs0, gh0 = cont(g_soft), cont(g_hard), cont(g_soft0), cont(g_hard0)
            check(lib.bg_attention_bwd_head_f32(ptr(g_out), _ld(g_out), ptr(x2), _ld(x2), ptr(soft0), ptr(soft),
                                                ptr(cv.rowptr), ptr(cv.col), ptr(gse), ptr(ghe), ptr(gs0), ptr(gh0),
                                                ctx.tau, ptr(gg), H, ptr(d_att), G_al.data_ptr() + 4 * H, H + 4, N, H,
                                                stream()), "bg_attention_bwd_head_f32")
            if ctx.needs_input_grad[1]:
                g_x = spmm_raw(plan.cross_v_sum.T, gg, soft)
            return AttentionNode._backward_tail(ctx, lib, G_al, d_att, g_out, g_x)
        if g_out is not None:
            g_out = _f32c(g_out)
            gathered = spmm_raw(plan.cross_v_sum, x, soft)
            g_more = sddmm_raw(_identity_op(plan), g_out, gathered)             # <g_out[v], gathered[v]>: to soft0
            gg = g_out * soft0                                                  # to gathered
            g_e = sddmm_raw(plan.cross_v_sum, gg, x)                            # to soft_e
            if ctx.needs_input_grad[1]:
                g_x = spmm_raw(plan.cross_v_sum.T, gg, soft)
        for extra in (g_soft, g_hard):                                          # hard passes its gradient straight to soft
            if extra is not None:
                g_e = extra.reshape(-1) if g_e is None else g_e + extra.reshape(-1)
        # ---- one [Nv, H+4] matrix of gradients at the stacked Linear's output: [d a | d logits | 0 0]
        G_al = torch.empty(N, H + 4, dtype=torch.float32, device=dev)
        d_att = None
        if g_e is not None:
            d_att = torch.empty_like(soft)
            check(lib.bg_seg_softmax_bwd_f32(ptr(soft), ptr(_f32c(g_e)), ptr(cv.rowptr), ctx.tau, ptr(d_att), cv.n_rows,
                                             stream()), "bg_seg_softmax_bwd_f32")
        gs0 = _f32c(g_soft0).contiguous() if g_soft0 is not None else None
        gh0 = _f32c(g_hard0).contiguous() if g_hard0 is not None else None
        check(lib.bg_mask_gumbel_bwd_f32(ptr(soft0), ptr(gs0), ptr(gh0), ptr(g_more), G_al.data_ptr() + 4 * H, H + 4, N,
                                         stream()), "bg_mask_gumbel_bwd_f32")
        return AttentionNode._backward_tail(ctx, lib, G_al, d_att, g_out, g_x)

    @staticmethod
    def _backward_tail(ctx, lib, G_al, d_att, g_out, g_x):
        """From d_att (gradient at the cross-edge logits; None = zero) and the mask slot of G_al to the gradients
        of v, NN_p(x), the stacked weight / bias and theta."""
        v, x, b, a, Wcat, theta, soft0, soft = ctx.saved_tensors
        plan = ctx.plan
        cv = plan.cross_v
        N, H = v.shape
        dev = v.device
        th = _f32c(theta.reshape(-1))
        db = dth = None
        if d_att is not None:
            nb = max(1, min(1024, (cv.n_rows + 3) // 4))
            dpre = torch.empty(cv.nnz, H, dtype=torch.float32, device=dev)
            dth_p = torch.empty(nb, H, dtype=torch.float32, device=dev)
            check(lib.bg_cross_logits_bwd_f32(ptr(a), _ld(a), ptr(b), _ld(b), ptr(th), ptr(cv.rowptr), ptr(cv.col),
                                              ptr(d_att), ptr(dpre), ptr(G_al), H + 4, ptr(dth_p), nb, cv.n_rows, H,
                                              stream()), "bg_cross_logits_bwd_f32")
            if ctx.needs_input_grad[2]:
                db = spmm_raw(plan.edge_program_gather.T, dpre)
            if ctx.needs_input_grad[5]:
                dth = _wgrad(ctx.shared, "theta", ctx.first, theta, lambda o, acc: colsum_raw(dth_p, o, acc))
                dth = dth.view(ctx.theta_shape) if dth is not None else None
        else:
            G_al[:, :H].zero_()
        sh, first = ctx.shared, ctx.first
        gW = _wgrad(sh, "Wcat", first, Wcat, lambda o, acc: gemm_tn_raw(G_al, [v], o, acc)) if ctx.needs_input_grad[3] else None
        gb = _wgrad(sh, "bcat", first, G_al[0], lambda o, acc: colsum_raw(G_al, o, acc)) if ctx.needs_input_grad[4] else None
        gv = gemm_raw([G_al], Wcat, False, False, residual=g_out) if ctx.needs_input_grad[0] else None
        return gv, g_x, db, gW, gb, dth, None, None, None, None, None, None


def block_pos_term(pdm4: torch.Tensor, Wc: torch.Tensor, bc: torch.Tensor) -> torch.Tensor:
    """[N, H]: pdm4 Wc[:, 2H:2H+4]^T + bc -- what the composed generator block adds in EVERY one of its calls
    (same weights, same positions): evaluated once per forward (no autograd graph: see GBlock)."""
    H = Wc.shape[0]
    with torch.no_grad():
        return gemm_raw([pdm4], Wc[:, 2 * H:2 * H + 4], False, True, bias=bc)


class GBlock(torch.autograd.Function):
    """v + block(v) of the generator's composed VoxelBlock (models.VoxelBlock.compose) as ONE autograd node:

        gm = A v;  story = mean_story(v);  term = story (Wu3 - Wu4)^T + building Wu4^T  (per story)
        y = leaky([v | gm | p] Wc^T + bc + term[story of the voxel]);  out = v + y

    Forward = the fused launches of the graph-free pass (gather-mean + story projection, building term,
    product with the gathered term in its epilogue) plus the residual add: 4 launches for 7 of the op-by-op
    path.  The hand-written backward needs 15 for its 21: one pass turns the incoming gradient into
    g + gx_self + A^T gx_nbr, a second adds the story / building means' share.  First-order autograd only
    (the generator is never differentiated twice)."""

    @staticmethod
    def forward(ctx, v, Wc, bc, Wsb, pdm4, plan, slope: float, shared=None, first=True, ones_pad=False, posb=None):
        """``ones_pad``: pdm4's last column is all ones and Wc's last column all zeros (the caller's promise), so
        the last column of Wc's gradient IS the bias gradient and no column sum is launched.
        ``posb`` = pdm4 Wc[:, 2H:]^T + bc (block_pos_term), the same for every call of a forward: the product
        then starts from it and runs over K = 2H (the gradients do not change: they are taken from the
        formula, not from how the forward evaluated it)."""
        H = v.shape[1]
        ctx.shared, ctx.first, ctx.ones_pad = shared, first, bool(ones_pad)
        gm, pq = gather_story_raw(plan.vox_mean, plan.story_seg, v, Wsb)
        term = spmm_add_raw(plan.story_building_mean, pq[:, H:], pq[:, :H])
        # one launch writes both y = leaky(..) (kept for the backward) and v + y (the result)
        y = torch.empty(v.shape[0], Wc.shape[0], dtype=torch.float32, device=v.device)
        if posb is not None:
            out = gemm_raw([v, gm], Wc[:, :2 * H], False, True, act=ACT_LEAKY, slope=slope, pre_add=term,
                           pre_rows=plan.story_gather.csr.col, acc_init=posb, residual=v, pre_residual_out=y)
        else:
            out = gemm_raw([v, gm, pdm4], Wc, False, True, bias=bc, act=ACT_LEAKY, slope=slope, pre_add=term,
                           pre_rows=plan.story_gather.csr.col, residual=v, pre_residual_out=y)
        ctx.plan, ctx.slope = plan, slope
        ctx.save_for_backward(v, gm, y, Wc, Wsb, pdm4)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        v, gm, y, Wc, Wsb, pdm4 = ctx.saved_tensors
        if ctx.shared is not None:
            ctx.shared.enter()
        plan = ctx.plan
        H = v.shape[1]
        g = _f32c(g)
        seg = plan.story_seg
        if seg.nnz == v.shape[0] and seg.count is not None and H % 4 == 0 and H <= 256:
            gz, gterm, story = story_bwd_raw(seg, g, y, v, ctx.slope)             # one launch for the three
        else:
            gz = torch.ops.aten.leaky_relu_backward(g, y, ctx.slope, True)       # y = leaky(z) has the sign of z
            gterm = spmm_raw(plan.story_gather.T, gz)                             # [S, H]: sum over each story's voxels
            story = None
        gWc = gbc = gWsb = gv = None
        sh, first = ctx.shared, ctx.first
        if ctx.needs_input_grad[1]:
            gWc = _wgrad(sh, "Wc", first, Wc, lambda o, acc: gemm_tn_raw(gz, [v, gm, pdm4], o, acc))
        if ctx.needs_input_grad[2]:
            if ctx.ones_pad and ctx.needs_input_grad[1]:
                gbc = gWc[:, -1] if gWc is not None else None                     # gz^T 1, summed over the calls with gWc
            else:
                gbc = _wgrad(sh, "bc", first, gz[0], lambda o, acc: colsum_raw(gz, o, acc))
        sbm = plan.story_building_mean                                            # symmetric: its own transpose
        if ctx.needs_input_grad[3]:
            if story is None:
                story = spmm_raw(plan.story_mean, v)
            sb = spmm_raw(sbm, story)
            gWsb = _wgrad(sh, "Wsb", first, Wsb, lambda o, acc: gemm_tn_raw(gterm, [story, sb], o, acc))
        if ctx.needs_input_grad[0]:
            gx = gemm_raw([gz], Wc[:, :2 * H], False, False)                      # [N, 2H]: to v | to A v
            gsb = gemm_raw([gterm], Wsb, False, False)                            # [S, 2H]: to story | to building
            g_story = spmm_add_raw(sbm, gsb[:, H:], gsb[:, :H])
            # g + gx_self + A^T gx_nbr + the voxel's share of its story's gradient (story_mean^T), one pass
            sm = plan.story_mean
            gv = spmm_add_raw(plan.vox_mean.T, gx[:, H:], g, gx[:, :H],
                              gather=(g_story, plan.story_gather.csr.col, sm.row_scale, sm.row_mode))
        return gv, gWc, gbc, gWsb, None, None, None, None, None, None, None


def attention_forward_raw(a, logits, b, x, v, theta, plan, exp_mask, exp_att, tau: float, wd=None, bd=None):
    """One Attention call without an autograd graph (see bg_attention_fwd_f32).  ``logits`` [Nv,2], or None with
    the composed decoder (wd [2,H], bd [2]) to apply in the kernel.
    Returns (soft0 [Nv,1], hard0 [Nv,1], soft [Ec], hard [Ec], new voxel features [Nv,H])."""
    lib = _lib.load()
    _lib.require_device(v, "attention_forward")
    cv = plan.cross_v
    a2, b2, x2, v2 = _f32c(a), _f32c(b), _f32c(x), _f32c(v)
    l2 = _f32c(logits) if logits is not None else None
    wd2 = _f32c(wd) if wd is not None else None
    bd2 = _f32c(bd).contiguous() if bd is not None else None
    th, em, ea = _f32c(theta.reshape(-1)), _f32c(exp_mask).contiguous(), _f32c(exp_att.reshape(-1))
    n, H = v2.shape
    dev = v.device
    soft0 = torch.empty(n, 1, dtype=torch.float32, device=dev)
    hard0 = torch.empty(n, 1, dtype=torch.float32, device=dev)
    soft = torch.empty(cv.nnz, dtype=torch.float32, device=dev)
    hard = torch.empty(cv.nnz, dtype=torch.float32, device=dev)
    out = torch.empty(n, H, dtype=torch.float32, device=dev)
    check(lib.bg_attention_fwd_f32(ptr(a2), _ld(a2), ptr(l2), _ld(l2) if l2 is not None else 0, ptr(b2), _ld(b2),
                                   ptr(x2), _ld(x2), ptr(v2), _ld(v2),
                                   ptr(th), ptr(cv.rowptr), ptr(cv.col), ptr(em), ptr(ea), tau, ptr(soft0), ptr(hard0),
                                   ptr(soft), ptr(hard), ptr(out), _ld(out), n, H,
                                   ptr(wd2), _ld(wd2) if wd2 is not None else 0, ptr(bd2), stream()), "bg_attention_fwd_f32")
    return soft0, hard0, soft, hard, out


def seg_argmax(logit: torch.Tensor, plan) -> torch.Tensor:
    lib = _lib.load()
    cv = plan.cross_v
    l2 = _f32c(logit.reshape(-1))
    arg = torch.empty(cv.n_rows, dtype=torch.int32, device=logit.device)
    check(lib.bg_seg_argmax_f32(ptr(l2), ptr(cv.rowptr), ptr(arg), cv.n_rows, stream()), "bg_seg_argmax_f32")
    return arg



# ------------------------------------------------------------------------------- critic-update glue
def gp_sample_raw(label: torch.Tensor, out: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """The gradient penalty's sample point (reference losses.py:76-79) in one launch; inputs are data."""
    lib = _lib.load()
    _lib.require_device(label, "gp_sample")
    l2, o2, u2 = _f32c(label), _f32c(out), _f32c(u.reshape(-1))
    n, Cn = l2.shape
    if tuple(o2.shape) != (n, Cn) or u2.numel() != n:
        raise _lib.BganError("gp_sample: label / out must be [n, C] and u must hold n weights")
    sample = torch.empty(n, Cn, dtype=torch.float32, device=label.device)
    check(lib.bg_gp_sample_f32(ptr(l2), _ld(l2), ptr(o2), _ld(o2), ptr(u2), ptr(sample), Cn, n, Cn, stream()),
          "bg_gp_sample_f32")
    return sample


def pack_columns_raw(srcs: Sequence[torch.Tensor], group: int, dst: torch.Tensor) -> torch.Tensor:
    """dst[:, p*group + j] = srcs[p][:, j] (zero for j >= C): label sets side by side in padded column groups."""
    lib = _lib.load()
    ss = [_f32c(t) for t in srcs]
    n, Cn = ss[0].shape
    if len(ss) > 4 or any(tuple(t.shape) != (n, Cn) for t in ss) or dst.shape[0] != n or dst.stride(-1) != 1:
        raise _lib.BganError("pack_columns: 1..4 sources of one shape, dst with matching rows")
    pa = _PtrArr4(*([t.data_ptr() for t in ss] + [None] * (4 - len(ss))))
    la = _I64Arr4(*([_ld(t) for t in ss] + [0] * (4 - len(ss))))
    check(lib.bg_pack_columns_f32(pa, la, len(ss), Cn, group, ptr(dst), dst.stride(0), n, stream()), "bg_pack_columns_f32")
    return dst


def construct_output_raw(mask_hard, att_hard, att_soft, plan, cls):
    """Attention.construct_output without an autograd graph: (out [Nv,C], soft_out [Nv,C], idx [Nv] int64)."""
    lib = _lib.load()
    _lib.require_device(mask_hard, "construct_output")
    cv = plan.cross_v
    mh, ah, asf, c2 = (_f32c(mask_hard.reshape(-1)), _f32c(att_hard.reshape(-1)), _f32c(att_soft.reshape(-1)), _f32c(cls))
    if mh.numel() != cv.n_rows or ah.numel() != cv.nnz or asf.numel() != cv.nnz or c2.shape[0] != cv.n_cols:
        raise _lib.BganError("construct_output: operand sizes do not match the plan")
    n, Cn = cv.n_rows, c2.shape[1]
    out = torch.empty(n, Cn, dtype=torch.float32, device=mh.device)
    soft = torch.empty(n, Cn, dtype=torch.float32, device=mh.device)
    idx = torch.empty(n, dtype=torch.int64, device=mh.device)
    check(lib.bg_construct_output_f32(ptr(mh), ptr(ah), ptr(asf), ptr(cv.rowptr), ptr(cv.col), ptr(c2), _ld(c2), Cn,
                                      ptr(out), ptr(soft), ptr(idx), n, stream()), "bg_construct_output_f32")
    return out, soft, idx


def pos_terms_raw(op: LinOp, pos: torch.Tensor, pad: float = 0.0) -> torch.Tensor:
    """[N, 4]: has * pos - mean over in-neighbours of pos, 4th column = ``pad`` (``pos`` may be a column slice)."""
    lib = _lib.load()
    _lib.require_device(pos, "pos_terms")
    c = op.csr
    p2 = pos if (pos.dtype == torch.float32 and pos.stride(-1) == 1) else pos.float().contiguous()
    if p2.shape[1] != 3 or p2.shape[0] != c.n_rows or c.n_rows != c.n_cols or c.col is None:
        raise _lib.BganError("pos_terms: pos must be [N, 3] over a square CSR with explicit columns")
    out = torch.empty(c.n_rows, 4, dtype=torch.float32, device=pos.device)
    check(lib.bg_pos_terms_f32(ptr(p2), p2.stride(0), ptr(c.rowptr), ptr(c.col), ptr(out), c.n_rows, float(pad), stream()),
          "bg_pos_terms_f32")
    return out


def d_static_cols_raw(feat, level, ML: int, pdm4, out, ds: int) -> None:
    lib = _lib.load()
    f2, p4 = _f32c(feat), _f32c(pdm4).contiguous()
    n, nf = f2.shape
    check(lib.bg_d_static_cols_f32(ptr(f2), _ld(f2), nf, ptr(level), ML, ptr(p4), ptr(out), out.stride(0), ds, n, stream()),
          "bg_d_static_cols_f32")


class CriticLoss(torch.autograd.Function):
    """(<val_b, wb> + <val_s, ws> + coef * sum_r (||grads_r|| - 1)^2,  sum_r (||grads_r|| - 1)^2) as one node (two
    launches forward, one backward); first-order autograd; the second output is for logging.  ``grads``
    [.., W] may be None (no penalty)."""

    @staticmethod
    def forward(ctx, val_b, val_s, grads, wb, ws, coef: float):
        lib = _lib.load()
        vb, vs = _f32c(val_b.reshape(-1)), _f32c(val_s.reshape(-1))
        if vb.numel() != wb.numel() or vs.numel() != ws.numel():
            raise _lib.BganError("critic_loss: weights must match the values")
        g2 = None
        if grads is not None:
            g2 = _f32c(grads.reshape(-1, grads.shape[-1]))
        dev = vb.device
        partial = torch.empty(128, dtype=torch.float32, device=dev)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        n_rows, W = (g2.shape if g2 is not None else (0, 0))
        check(lib.bg_critic_loss_f32(ptr(vb), ptr(wb), vb.numel(), ptr(vs), ptr(ws), vs.numel(), ptr(g2),
                                     _ld(g2) if g2 is not None else 0, n_rows, W, float(coef), ptr(partial), ptr(loss),
                                     stream()), "bg_critic_loss_f32")
        ctx.coef = float(coef)
        ctx.shapes = (val_b.shape, val_s.shape, None if grads is None else grads.shape)
        ctx.save_for_backward(wb, ws, g2)
        total, pen = loss[0], loss[1]
        ctx.mark_non_differentiable(pen)
        return total, pen

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go, _gpen):
        lib = _lib.load()
        wb, ws, g2 = ctx.saved_tensors
        sb, ss, sg = ctx.shapes
        dev = wb.device
        go = _f32c(go.reshape(1))
        db = torch.empty(sb, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        ds = torch.empty(ss, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        dg = torch.empty(sg, dtype=torch.float32, device=dev) if (g2 is not None and ctx.needs_input_grad[2]) else None
        n_rows, W = (g2.shape if g2 is not None else (0, 0))
        check(lib.bg_critic_loss_bwd_f32(ptr(go), ptr(wb), wb.numel(), ptr(ws), ws.numel(), ptr(g2),
                                         _ld(g2) if g2 is not None else 0, n_rows, W, ctx.coef, ptr(db), ptr(ds), ptr(dg),
                                         stream()), "bg_critic_loss_bwd_f32")
        return db, ds, dg, None, None, None


def _critic_head_struct(plan, N, L, ds, dl, H2, Hd, Zs, Zk, W, dec_b, dec_s, wb, ws):
    """include/bgan.h bgCriticHead for a plan and the given operands (raw addresses: the caller keeps them alive)."""
    blocks, st, sg = plan.vox_blocks, plan.story_seg, plan.story_graph_seg
    A, AT = plan.vox_mean, plan.vox_mean.T
    if blocks is None or st.seg_of is None or sg.seg_of is None or plan.graph_seg.seg_of is None:
        raise _lib.BganError("critic_head: the plan lacks diagonal blocks / segment tables")
    return _lib.CriticHead(N, plan.S, plan.B, L, ds, dl, H2, Hd, blocks.n, blocks.max_rows, blocks.max_nnz, blocks.max_deg,
                           D_BASIS,
                           ptr(Zs), ptr(Zk), ptr(W), *[ptr(t) for t in dec_b], *[ptr(t) for t in dec_s], ptr(wb), ptr(ws),
                           ptr(st.rowptr), ptr(st.col), ptr(st.seg_of), ptr(sg.rowptr), ptr(sg.col), ptr(sg.seg_of),
                           ptr(plan.graph_seg.seg_of), ptr(A.csr.rowptr), ptr(A.csr.col), ptr(A.row_scale),
                           ptr(AT.csr.rowptr), ptr(AT.csr.col), ptr(blocks.ptr))


class GLossHead(torch.autograd.Function):
    """The generator's adversarial objective through the frozen collapsed discriminator,
    <D_building(label), wb> + <D_story(label), ws>, as ONE host call each way (bg_gloss_head_forward_f32 / _backward_f32):
    the only differentiable input is ``label`` [N, n_lab] (the generated labels); W, the decoders, Zs are data here."""

    @staticmethod
    def forward(ctx, label, W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2, Zs, plan, wb, ws, L: int, ds: int, dl: int):
        lib = _lib.load()
        _lib.require_device(label, "gloss_head")
        label = _f32c(label)
        ts = (W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2, Zs, wb, ws)
        if any(t.dtype != torch.float32 or not t.is_contiguous() for t in ts):
            raise _lib.BganError("gloss_head: contiguous fp32 operands")
        N, n_lab = label.shape
        H2, Hd = W.shape[1], bw1.shape[0]
        if (N != plan.Nv or tuple(W.shape) != ((L + 1) * (ds + dl), H2) or tuple(Zs.shape) != (N, (L + 1) * ds) or n_lab > dl
                or wb.numel() < plan.B or ws.numel() < plan.S):
            raise _lib.BganError("gloss_head: operand shapes do not fit the plan / the coefficient table")
        h = _critic_head_struct(plan, N, L, ds, dl, H2, Hd, Zs, None, W, (bw1, bb1, bw2, bb2), (sw1, sb1, sw2, sb2), wb, ws)
        n_ws = lib.bg_gloss_head_workspace(C.byref(h))
        if n_ws < 0:
            raise _lib.BganError("gloss_head: sizes not supported")
        work = torch.empty(n_ws, dtype=torch.float32, device=label.device)
        loss = torch.empty(2, dtype=torch.float32, device=label.device)
        check(lib.bg_gloss_head_forward_f32(C.byref(h), ptr(label), _ld(label), n_lab, ptr(work), n_ws, ptr(loss), stream()),
              "bg_gloss_head_forward_f32")
        ctx.h, ctx.work, ctx.n_ws, ctx.shape = h, work, n_ws, (N, n_lab)
        ctx.keep = (Zs, wb, ws, plan)
        ctx.save_for_backward(W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2)
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        lib = _lib.load()
        go = _f32c(go.reshape(1))
        N, n_lab = ctx.shape
        g = torch.empty(N, n_lab, dtype=torch.float32, device=go.device)
        check(lib.bg_gloss_head_backward_f32(C.byref(ctx.h), n_lab, ptr(go), ptr(ctx.work), ctx.n_ws, ptr(g), stream()),
              "bg_gloss_head_backward_f32")
        ctx.work = None
        return (g,) + (None,) * 16


class ProgramStack(torch.autograd.Function):
    """ProgramGNN.forward -- encoder and every ProgramBlock call with its residual -- as ONE autograd node whose
    forward and backward are one host call each (bg_pstack_forward_f32 / _backward_f32).  First-order autograd.
    Differentiable inputs: enc_w [H, kin], enc_b, Wm [H, 2H], bm, Wu [H, 3H], bu; program_in [Np, kin] is data."""

    @staticmethod
    def forward(ctx, enc_w, enc_b, Wm, bm, Wu, bu, program_in, plan, L: int):
        lib = _lib.load()
        _lib.require_device(program_in, "program_stack")
        ts = [t.contiguous() for t in (enc_w, enc_b, Wm, bm, Wu, bu)]
        program_in = _f32c(program_in).contiguous()
        H, kin = ts[0].shape
        Np = program_in.shape[0]
        if program_in.shape[1] != kin or Np != plan.Np or tuple(ts[2].shape) != (H, 2 * H) or tuple(ts[4].shape) != (H, 3 * H):
            raise _lib.BganError("program_stack: operand shapes do not fit")
        ps = plan.ops.get("_generator_plan_struct")
        if ps is None:
            raise _lib.BganError("program_stack: the plan carries no bgGeneratorPlan (models._generator_plan_struct)")
        ws = _lib.GeneratorWeights(H, 0, L, 0, kin - 1, 0, ptr(ts[0]), ptr(ts[1]), ptr(ts[2]), ptr(ts[3]), ptr(ts[4]), ptr(ts[5]),
                                   None, None, None, None, 0, None, None, None, None, None, None, None)
        n_ws = int(lib.bg_pstack_workspace(Np, H, kin, L))
        if n_ws < 0:
            raise _lib.BganError("program_stack: sizes not supported")
        work = torch.empty(n_ws, dtype=torch.float32, device=program_in.device)
        x = torch.empty(Np, H, dtype=torch.float32, device=program_in.device)
        check(lib.bg_pstack_forward_f32(C.byref(ws), C.byref(ps), ptr(program_in), ptr(work), n_ws, ptr(x), stream()),
              "bg_pstack_forward_f32")
        ctx.ws, ctx.ps, ctx.work, ctx.n_ws, ctx.plan = ws, ps, work, n_ws, plan
        ctx.save_for_backward(program_in, *ts)
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        program_in, *ts = ctx.saved_tensors
        plan = ctx.plan
        pr, cnt = plan.prog_mean.T.csr, plan.prog_mean.row_scale
        g = _f32c(g).contiguous()
        outs = [torch.empty_like(t) for t in ts]
        check(lib.bg_pstack_backward_f32(C.byref(ctx.ws), C.byref(ctx.ps), ptr(pr.rowptr), ptr(pr.col), ptr(cnt),
                                         ptr(program_in), ptr(g), ptr(ctx.work), ctx.n_ws, *[ptr(t) for t in outs], stream()),
              "bg_pstack_backward_f32")
        ctx.work = None
        return (*outs, None, None, None)


class VoxelStackG(torch.autograd.Function):
    """VoxelGNN_G.forward below the program GNN -- encoder, NN_p(x), every Attention and VoxelBlock call -- as ONE
    autograd node whose forward and backward are one host call each (bg_gstack_forward_f32 / _backward_f32: the
    launches of ops.AttentionNode / ops.GBlock and their hand-written backward, issued from C++; every call's input,
    neighbour mean and activation stay in one workspace).  First-order autograd.

    Differentiable inputs: x [Np, H], enc_w [H, 4], enc_b, nnp_w, nnp_b, Wcat [H+4, H], bcat, theta, Wc [H, 2H+4],
    bc, Wsb.  Returns the last Attention call's (mask soft [Nv,1], mask hard, pointer soft [Ec], pointer hard)."""

    @staticmethod
    def forward(ctx, x, enc_w, enc_b, nnp_w, nnp_b, Wcat, bcat, theta, Wc, bc, Wsb, pe, v4, pdm4, noise, plan, L: int):
        lib = _lib.load()
        _lib.require_device(x, "voxel_stack")
        H = x.shape[1]
        Np, Nv, S, Ec = plan.Np, plan.Nv, plan.S, plan.Ec
        dev = x.device
        x, v4, pdm4 = _f32c(x).contiguous(), _f32c(v4).contiguous(), _f32c(pdm4).contiguous()
        ts = [enc_w, enc_b, nnp_w, nnp_b, Wcat, bcat, theta, bc, Wsb, pe]
        ts = [t.contiguous() for t in ts]
        enc_w, enc_b, nnp_w, nnp_b, Wcat, bcat, theta, bc, Wsb, pe = ts
        if Wc.stride(1) != 1:
            Wc = Wc.contiguous()
        n_calls = 1 + sum(1 for i in range(L) if i % 2 == 0 and i != 0)
        keep = [(_f32c(a).contiguous(), _f32c(b.reshape(-1)).contiguous()) for a, b in list(noise)[:n_calls]]
        if len(keep) != n_calls or any(tuple(a.shape) != (Nv, 2) or b.numel() != Ec for a, b in keep):
            raise _lib.BganError("voxel_stack: noise must hold one ([Nv, 2], [Ec]) pair per Attention call")
        PA = _lib.ptr_array_type(n_calls)
        em, ea = PA(*[a.data_ptr() for a, _ in keep]), PA(*[b.data_ptr() for _, b in keep])
        ws = _lib.GeneratorWeights(H, 0, 0, L, 1, pe.shape[0], None, None, None, None, None, None, ptr(enc_w), ptr(enc_b),
                                   ptr(pe), ptr(Wc), Wc.stride(0), ptr(bc), ptr(Wsb), ptr(Wcat), ptr(bcat), ptr(nnp_w),
                                   ptr(nnp_b), ptr(theta))
        ps = plan.ops.get("_generator_plan_struct")
        if ps is None:
            raise _lib.BganError("voxel_stack: the plan carries no bgGeneratorPlan (models._generator_plan_struct)")
        n_ws = int(lib.bg_gstack_workspace(Np, Nv, S, Ec, H, L))
        if n_ws < 0:
            raise _lib.BganError("voxel_stack: sizes not supported")
        work = torch.empty(n_ws, dtype=torch.float32, device=dev)
        mask_hard = torch.empty(Nv, 1, dtype=torch.float32, device=dev)
        mask_soft = torch.empty(Nv, 1, dtype=torch.float32, device=dev)
        att_hard = torch.empty(Ec, dtype=torch.float32, device=dev)
        att_soft = torch.empty(Ec, dtype=torch.float32, device=dev)
        check(lib.bg_gstack_forward_f32(C.byref(ws), C.byref(ps), ptr(x), ptr(v4), ptr(pdm4), em, ea, n_calls, ptr(work), n_ws,
                                        ptr(mask_hard), ptr(mask_soft), ptr(att_hard), ptr(att_soft), stream()),
              "bg_gstack_forward_f32")
        ctx.ws, ctx.ps, ctx.work, ctx.n_ws, ctx.n_calls, ctx.plan = ws, ps, work, n_ws, n_calls, plan
        # (outputs go through save_for_backward: kept as plain attributes they would tie the node to its own outputs
        # in a reference cycle, and the dead graph -- with the parameters' AccumulateGrad nodes -- would linger until
        # the cycle collector runs, which breaks a later stream capture)
        ctx.save_for_backward(x, v4, pdm4, *ts, Wc, mask_soft, att_soft)
        ctx.set_materialize_grads(False)
        return mask_soft, mask_hard, att_soft, att_hard

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_ms, g_mh, g_as, g_ah):
        lib = _lib.load()
        x, v4, pdm4, enc_w, enc_b, nnp_w, nnp_b, Wcat, bcat, theta, bc, Wsb, pe, Wc, mask_soft, att_soft = ctx.saved_tensors
        plan = ctx.plan
        H = x.shape[1]
        dev = x.device
        cp, vt = plan.cross_p, plan.vox_mean.T.csr
        bp = _lib.GeneratorBackPlan(ptr(vt.rowptr), ptr(vt.col), ptr(cp.rowptr), ptr(cp.col), ptr(cp.perm),
                                    1 if cp.row_per_workgroup else 0)
        cont = lambda t: _f32c(t).reshape(-1).contiguous() if t is not None else None
        g_ms, g_mh, g_as, g_ah = cont(g_ms), cont(g_mh), cont(g_as), cont(g_ah)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        g_x, g_enc_w, g_enc_b, g_nnp_w, g_nnp_b = new(x.shape), new(enc_w.shape), new(enc_b.shape), new(nnp_w.shape), new(nnp_b.shape)
        g_Wcat, g_bcat, g_theta, g_Wc, g_Wsb = new(Wcat.shape), new(bcat.shape), new(theta.shape), new(H, 2 * H + 4), new(Wsb.shape)
        check(lib.bg_gstack_backward_f32(C.byref(ctx.ws), C.byref(ctx.ps), C.byref(bp), ptr(x), ptr(v4), ptr(pdm4), ctx.n_calls,
                                         ptr(ctx.work), ctx.n_ws, ptr(mask_soft), ptr(att_soft), ptr(g_mh), ptr(g_ms),
                                         ptr(g_ah), ptr(g_as), ptr(g_x), ptr(g_enc_w), ptr(g_enc_b), ptr(g_nnp_w),
                                         ptr(g_nnp_b), ptr(g_Wcat), ptr(g_bcat), ptr(g_theta), ptr(g_Wc), ptr(g_Wsb),
                                         stream()), "bg_gstack_backward_f32")
        ctx.work = None
        return (g_x, g_enc_w, g_enc_b, g_nnp_w, g_nnp_b, g_Wcat, g_bcat, g_theta, g_Wc, g_Wc[:, -1], g_Wsb,
                None, None, None, None, None, None)


class CriticHead(torch.autograd.Function):
    """One WGAN-GP critic update below the collapsed GNN's coefficient table as ONE host call each way
    (bg_critic_head_forward_f32 / _backward_f32: what DiscriminatorVoxel.critic_terms + CriticLoss launch op by op --
    thin products, pooling, decoders, closed-form penalty gradients, objective -- and their hand-derived backward).
    Inputs that get gradients: the coefficient table W and the two decoders' parameters; labels and operator powers
    are data.  Returns (loss, bare penalty sum) like CriticLoss; first-order autograd."""

    @staticmethod
    def forward(ctx, W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2, Zs, Zk, plan, wb, ws, coef: float, L: int, ds: int, dl: int):
        lib = _lib.load()
        _lib.require_device(W, "critic_head")
        ts = (W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2, Zs, Zk, wb, ws)
        if any(t.dtype != torch.float32 or not t.is_contiguous() for t in ts):
            raise _lib.BganError("critic_head: contiguous fp32 operands")
        N, H2, Hd = Zs.shape[0], W.shape[1], bw1.shape[0]
        K1 = L + 1
        if (tuple(W.shape) != (K1 * (ds + dl), H2) or tuple(Zs.shape) != (N, K1 * ds) or Zk.numel() != N * K1 * 3 * dl
                or tuple(bw1.shape) != (Hd, H2) or tuple(sw1.shape) != (Hd, H2) or bw2.numel() != Hd or sw2.numel() != Hd
                or bb1.numel() != Hd or sb1.numel() != Hd or bb2.numel() != 1 or sb2.numel() != 1
                or wb.numel() != 2 * plan.B or ws.numel() != 2 * plan.S or N != plan.Nv):
            raise _lib.BganError("critic_head: operand shapes do not fit the plan / the coefficient table")
        h = _critic_head_struct(plan, N, L, ds, dl, H2, Hd, Zs, Zk, W, (bw1, bb1, bw2, bb2), (sw1, sb1, sw2, sb2), wb, ws)
        n_ws = lib.bg_critic_head_workspace(C.byref(h))
        if n_ws < 0:
            raise _lib.BganError("critic_head: sizes not supported")
        work = torch.empty(n_ws, dtype=torch.float32, device=W.device)
        loss = torch.empty(2, dtype=torch.float32, device=W.device)
        check(lib.bg_critic_head_forward_f32(C.byref(h), float(coef), ptr(work), n_ws, ptr(loss), stream()),
              "bg_critic_head_forward_f32")
        ctx.h, ctx.work, ctx.n_ws, ctx.coef = h, work, n_ws, float(coef)
        ctx.keep = (Zs, Zk, wb, ws, plan)                       # the struct holds raw addresses
        ctx.save_for_backward(W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2)
        total, pen = loss[0], loss[1]
        ctx.mark_non_differentiable(pen)
        return total, pen

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go, _gpen):
        lib = _lib.load()
        W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2 = ctx.saved_tensors
        go = _f32c(go.reshape(1))
        outs = [torch.empty_like(t) for t in (W, bw1, bb1, bw2, bb2, sw1, sb1, sw2, sb2)]
        check(lib.bg_critic_head_backward_f32(C.byref(ctx.h), ctx.coef, ptr(go), ptr(ctx.work), ctx.n_ws,
                                              *[ptr(t) for t in outs], stream()), "bg_critic_head_backward_f32")
        ctx.work = None
        return (*outs, None, None, None, None, None, None, None, None, None)


def adam_step(param, grad, m, v, lr, beta1, beta2, eps, step, step_dev=None):
    """``step`` (host int) or ``step_dev`` (int32 device tensor holding the step count)."""
    lib = _lib.load()
    check(lib.bg_adam_f32(ptr(param), ptr(grad), ptr(m), ptr(v), param.numel(), lr, beta1, beta2, eps,
                          int(step) if step_dev is None else 0, ptr(step_dev), stream()), "bg_adam_f32")


# ------------------------------------------------------------------------------- fused layer stack
class AffineStack(torch.autograd.Function):
    """L residual message-passing layers with ONE shared affine map (the discriminator's composed
    VoxelBlock, models.VoxelBlock.compose):

        v_{l+1} = v_l + [v_l | A v_l | p] Wc^T + bc ,   l = 0..L-1 ,   A = the plan's gather-mean.

    One autograd node for the whole stack: layer inputs live in one [L, N, H] buffer, so the weight
    gradient of all L layers is a single split-K GEMM over L*N rows, the bias gradient one column
    sum, and each layer's input gradient  g + gx[:, :H] + A^T gx[:, H:2H]  one fused gather pass.
    Differentiable twice: under create_graph the input gradient is produced by AffineStackT, whose
    own backward is hand-written (third order is not provided).
    """

    @staticmethod
    def forward(ctx, v0, Wc, bc, p, op: LinOp, L: int):
        N, H = v0.shape
        dev = v0.device
        Vs = torch.empty(L + 1, N, H, dtype=torch.float32, device=dev)
        GMs = torch.empty(L, N, H, dtype=torch.float32, device=dev)
        Vs[0].copy_(v0)
        for l in range(L):
            spmm_raw(op, Vs[l], out=GMs[l])
            gemm_raw([Vs[l], GMs[l], p], Wc, False, True, bias=bc, residual=Vs[l], out=Vs[l + 1])
        ctx.op, ctx.L = op, L
        ctx.inputs_only = _INPUT_GRADS_ONLY
        ctx.save_for_backward(Vs, GMs, Wc, p)
        return Vs[L]

    @staticmethod
    def backward(ctx, g):
        Vs, GMs, Wc, p = ctx.saved_tensors
        op, L = ctx.op, ctx.L
        N, H = g.shape
        need_w = ctx.needs_input_grad[1] and not ctx.inputs_only
        need_b = ctx.needs_input_grad[2] and not ctx.inputs_only
        if torch.is_grad_enabled():
            # create_graph: the input gradient must itself be differentiable
            gv = AffineStackT.apply(g, Wc, op, L) if ctx.needs_input_grad[0] else None
            gW = gb = None
            if need_w or need_b:
                raise NotImplementedError("AffineStack: differentiable weight gradients are only provided through "
                                          "ops.input_grads_only() (use COMPOSE_STACK = False for the generic path)")
            return gv, gW, gb, None, None, None
        Gs, g0 = _stack_transpose_chain(g, Wc, op, L, keep=need_w or need_b)
        gW = gb = None
        if need_w:
            prep = p.unsqueeze(0).expand(L, N, p.shape[1]).reshape(L * N, p.shape[1])
            gW = gemm_tn_raw(Gs.view(L * N, H), [Vs[:L].reshape(L * N, H), GMs.view(L * N, H), prep])
        if need_b:
            gb = colsum_raw(Gs.view(L * N, H))
        return (g0 if ctx.needs_input_grad[0] else None), gW, gb, None, None, None


def _stack_transpose_chain(g, Wc, op: LinOp, L: int, keep: bool):
    """G_L = g;  G_l = G_{l+1} + X[:, :H] + A^T X[:, H:2H],  X = G_{l+1} Wc.  Returns (Gs[l] = G_{l+1}, G_0)."""
    N, H = g.shape
    g = _f32c(g)
    Gs = torch.empty(L, N, H, dtype=torch.float32, device=g.device) if keep else None
    X = torch.empty(N, Wc.shape[1], dtype=torch.float32, device=g.device)
    cur = g
    if keep:
        Gs[L - 1].copy_(g)
        cur = Gs[L - 1]
    for l in range(L - 1, -1, -1):
        gemm_raw([cur], Wc, False, False, out=X)
        if keep and l > 0:
            nxt = Gs[l - 1]
        else:
            nxt = torch.empty(N, H, dtype=torch.float32, device=g.device)
        spmm_add_raw(op.T, X[:, H:2 * H], cur, X[:, :H], out=nxt)
        cur = nxt
    return Gs, cur


class AffineStackT(torch.autograd.Function):
    """g0 = T^L g with T(G) = G + (G Wc)[:, :H] + A^T (G Wc)[:, H:2H]: the input gradient of AffineStack
    as a differentiable function of (g, Wc).  Its backward (the WGAN-GP double backward) is
        U_{l+1} = U_l + [U_l | A U_l] Wc[:, :2H]^T ,   dWc[:, :2H] = sum_l G_{l+1}^T [U_l | A U_l]."""

    @staticmethod
    def forward(ctx, g, Wc, op: LinOp, L: int):
        Gs, g0 = _stack_transpose_chain(g, Wc, op, L, keep=True)
        ctx.op, ctx.L = op, L
        ctx.save_for_backward(Gs, Wc)
        return g0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u):
        Gs, Wc = ctx.saved_tensors
        op, L = ctx.op, ctx.L
        N, H = u.shape
        dev = u.device
        Us = torch.empty(L + 1, N, H, dtype=torch.float32, device=dev)
        AUs = torch.empty(L, N, H, dtype=torch.float32, device=dev)
        Us[0].copy_(u)
        W2 = Wc[:, :2 * H]
        for l in range(L):
            spmm_raw(op, Us[l], out=AUs[l])
            gemm_raw([Us[l], AUs[l]], W2, False, True, residual=Us[l], out=Us[l + 1])
        gW = None
        if ctx.needs_input_grad[1]:
            # G_{l+1} pairs with U_l:  Gs[l] holds G_{l+1}
            gW2 = gemm_tn_raw(Gs.view(L * N, H), [Us[:L].reshape(L * N, H), AUs.view(L * N, H)])
            gW = torch.nn.functional.pad(gW2, (0, Wc.shape[1] - 2 * H))
        return (Us[L] if ctx.needs_input_grad[0] else None), gW, None, None


def cheb_shift_raw(X: torch.Tensor, out: torch.Tensor, K: int, K_out: int, block_floats: int, transpose: bool) -> None:
    """include/bgan.h bg_cheb_shift_f32: multiplication by x on K blocks of Chebyshev coefficients (or its adjoint)."""
    lib = _lib.load()
    _lib.require_device(X, "cheb_shift")
    if (not X.is_contiguous() or not out.is_contiguous() or X.numel() < K * block_floats
            or out.numel() < K_out * block_floats or X.dtype != torch.float32 or out.dtype != torch.float32):
        raise _lib.BganError("cheb_shift: contiguous fp32 tables of K / K_out blocks expected")
    check(lib.bg_cheb_shift_f32(ptr(X), ptr(out), K, K_out, block_floats, 1 if transpose else 0, stream()),
          "bg_cheb_shift_f32")


def _coef_shifts(Xall: torch.Tensor, L: int, du: int, H2: int) -> torch.Tensor:
    """shift(X^m) of every layer's state by one launch: [L S + du, H2] laid out like Xall's first L slabs."""
    lib = _lib.load()
    S = (L + 2) * du
    out = torch.empty(L * S + du, H2, dtype=torch.float32, device=Xall.device)
    check(lib.bg_cheb_shift_slabs_f32(ptr(Xall), ptr(out), L * (L + 2) + 1, L + 2, 1, L + 1, du * H2, 0, stream()),
          "bg_cheb_shift_slabs_f32")
    return out


class PolyCoef(torch.autograd.Function):
    """Coefficients of the collapsed discriminator GNN (models.VoxelGNN_D.forward_collapsed).

    State X^m [(L+1)*du, H]: row block k holds T^m_k (du x H), the map from the du input columns to
    the hidden state through k applications of the aggregation operator.  One layer is
        X^{m+1} = X^m + [X^m | shift_du(X^m)] Wc[:, :2H]^T + [E; 0; ..]       (shift = one block down)
    i.e. ONE two-segment GEMM whose second A segment is the first at a row offset (the buffers carry
    du zero rows above block 0).  Returns X^L.  The backward runs the transposed recurrence and
    gets dWc[:, :2H] of all L layers from a single split-K GEMM over the stacked slabs.
    """

    @staticmethod
    def forward(ctx, Wc, X0, E, L: int, basis=None):
        du, H = X0.shape
        basis = D_BASIS if basis is None else basis
        N = (L + 1) * du
        S = N + du                                    # slab = du zero rows + the state
        dev = X0.device
        Xall = torch.zeros(L + 1, S, H, dtype=torch.float32, device=dev)
        Xall[0, du:2 * du].copy_(X0)
        Efull = torch.zeros(N, H, dtype=torch.float32, device=dev)
        Efull[:du].copy_(E)
        # Chebyshev basis: shift(X^m) = x * X^m (cheb_shift_raw), kept per layer for the backward at the state's rows
        ShAll = torch.zeros(L * S + du, H, dtype=torch.float32, device=dev) if basis == CHAIN_CHEB else None
        W2 = Wc[:, :2 * H]
        for m in range(L):
            rows = min(N, (m + 2) * du)               # blocks above m+1 are still zero
            cur = Xall[m]
            shifted = cur[:rows]                      # monomials: one block down = a row offset
            if basis == CHAIN_CHEB:
                shifted = ShAll[m * S + du:m * S + du + rows]
                cheb_shift_raw(cur[du:], shifted, L + 1, rows // du, du * H, False)
            gemm_raw([cur[du:du + rows], shifted], W2, False, True, residual=cur[du:du + rows],
                     pre_add=Efull[:rows], out=Xall[m + 1, du:du + rows])
        ctx.L, ctx.du, ctx.basis = L, du, basis
        ctx.save_for_backward(Xall, Wc, ShAll)
        return Xall[L, du:]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        Xall, Wc, ShAll = ctx.saved_tensors
        L, du, basis = ctx.L, ctx.du, ctx.basis
        H = g.shape[1]
        N = (L + 1) * du
        S = N + du
        dev = g.device
        # G^m at rows [0, N) of slab m, du zero rows below it (the shift-up view reads them)
        Gall = torch.zeros(L + 1, S, H, dtype=torch.float32, device=dev)
        Gall[L, :N].copy_(g)
        Wab = torch.cat([Wc[:, :H], Wc[:, H:2 * H]], dim=0)          # [2H, H]: G Wa + shiftup(G) Wb
        sht = torch.empty(N, H, dtype=torch.float32, device=dev) if basis == CHAIN_CHEB else None
        for m in range(L - 1, -1, -1):
            nxt = Gall[m + 1]
            up = nxt[du:du + N]
            if basis == CHAIN_CHEB:                                   # the adjoint of x * (.)
                cheb_shift_raw(nxt, sht, L + 1, L + 1, du * H, True)
                up = sht
            gemm_raw([nxt[:N], up], Wab, False, False, residual=nxt[:N], out=Gall[m, :N])
        gW = gX0 = gE = None
        if ctx.needs_input_grad[0]:
            # dWc[:, :2H] = sum_m G^{m+1 T} [X^m | shift(X^m)]: slabs share one pitch, so the sum over
            # layers is one long reduction (pad rows of G are zero, whatever X rows they meet)
            Gf = Gall[1:L + 1].reshape(L * S, H)
            Xf = Xall.view((L + 1) * S, H)
            gW2 = gemm_tn_raw(Gf, [Xf[du:du + L * S], ShAll[du:du + L * S] if basis == CHAIN_CHEB else Xf[:L * S]])
            gW = torch.nn.functional.pad(gW2, (0, Wc.shape[1] - 2 * H))
        if ctx.needs_input_grad[1]:
            gX0 = Gall[0, :du].clone()
        if ctx.needs_input_grad[2]:
            gE = Gall[1:L + 1, :du].sum(dim=0)
        return gW, gX0, gE, None, None


class PolyCoefD(torch.autograd.Function):
    """PolyCoef with its inputs built from the discriminator's parameters inside the node: X0 (how the input
    columns enter v_0 through feature_enc / pe / label_enc) and E (the position block of Wc and the bias, added
    by every layer) are written straight into the recurrence's buffers by one launch (bg_d_coef_init_f32), and
    the backward hands the parameters their gradients from one launch (bg_d_coef_init_bwd_f32) -- instead of
    ~10 cat / transpose / copy launches forward and as many slice / add launches backward.

    Wc [2H, 4H+4], bc [2H]; fe_w [H, nf], fe_b [H]; le_w [H, nl], le_b [H]; pe [ML, H] (no gradient).
    Returns X^L [(L+1)*(ds+dl), 2H] as PolyCoef does."""

    @staticmethod
    def forward(ctx, Wc, bc, fe_w, fe_b, le_w, le_b, pe, L: int, ds: int, dl: int, basis=None):
        lib = _lib.load()
        _lib.require_device(Wc, "PolyCoefD")
        H2 = Wc.shape[0]
        H, nf, nl, ML = H2 // 2, fe_w.shape[1], le_w.shape[1], pe.shape[0]
        if (H2 != 2 * H or Wc.shape[1] != 2 * H2 + 4 or fe_w.shape[0] != H or le_w.shape[0] != H or pe.shape[1] != H
                or ds < nf + ML + 5 or dl < nl):
            raise _lib.BganError("PolyCoefD: parameter shapes do not fit the collapsed discriminator")
        du = ds + dl
        N = (L + 1) * du
        S = N + du
        dev = Wc.device
        Wc2, args = _f32c(Wc), [_f32c(t).contiguous() for t in (fe_w, fe_b, le_w, le_b, pe, bc)]
        basis = D_BASIS if basis is None else basis
        cheb = basis == CHAIN_CHEB
        n_x, n_e = (L + 1) * S * H2, N * H2
        n_sh = (L * S + du) * H2 if (cheb and not FUSE_COEF_RECURRENCE) else 0       # the one-call form keeps no shifts
        if FUSE_COEF_RECURRENCE:
            # clear + parameter rows + the L products in one host call (shapes depend on the parameters alone)
            buf = torch.empty(n_x + n_e + n_sh, dtype=torch.float32, device=dev)
        else:
            buf = torch.zeros(n_x + n_e + n_sh, dtype=torch.float32, device=dev)     # one fill for all three
        Xall = buf[:n_x].view(L + 1, S, H2)
        Efull = buf[n_x:n_x + n_e].view(N, H2)
        ShAll = buf[n_x + n_e:].view(L * S + du, H2) if n_sh else None
        if FUSE_COEF_RECURRENCE:
            check(lib.bg_d_coef_forward_f32(ptr(args[0]), ptr(args[1]), ptr(args[2]), ptr(args[3]), ptr(args[4]), ptr(Wc2),
                                            _ld(Wc2), ptr(args[5]), H, nf, nl, ML, ds, dl, L, basis, ptr(Xall), ptr(Efull),
                                            stream()), "bg_d_coef_forward_f32")
        else:
            check(lib.bg_d_coef_init_f32(ptr(args[0]), ptr(args[1]), ptr(args[2]), ptr(args[3]), ptr(args[4]), ptr(Wc2),
                                         _ld(Wc2), ptr(args[5]), H, nf, nl, ML, ds, dl, ptr(Xall[0, du:2 * du]),
                                         ptr(Efull[:du]), stream()), "bg_d_coef_init_f32")
            W2 = Wc2[:, :2 * H2]
            for m in range(L):
                rows = min(N, (m + 2) * du)               # blocks above m+1 are still zero
                cur = Xall[m]
                shifted = cur[:rows]
                if cheb:
                    shifted = ShAll[m * S + du:m * S + du + rows]
                    cheb_shift_raw(cur[du:], shifted, L + 1, rows // du, du * H2, False)
                gemm_raw([cur[du:du + rows], shifted], W2, False, True, residual=cur[du:du + rows],
                         pre_add=Efull[:rows], out=Xall[m + 1, du:du + rows])
        ctx.dims = (L, du, ds, dl, H, nf, nl, ML, basis)
        ctx.save_for_backward(Xall, Wc2, ShAll)
        return Xall[L, du:]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        Xall, Wc, ShAll = ctx.saved_tensors
        L, du, ds, dl, H, nf, nl, ML, basis = ctx.dims
        cheb = basis == CHAIN_CHEB
        H2 = 2 * H
        N = (L + 1) * du
        S = N + du
        dev = g.device
        gW = torch.empty(H2, 2 * H2 + 4, dtype=torch.float32, device=dev)
        g_fe_w = torch.empty(H, nf, dtype=torch.float32, device=dev)
        g_le_w = torch.empty(H, nl, dtype=torch.float32, device=dev)
        g_fe_b = torch.empty(H, dtype=torch.float32, device=dev)
        g_le_b = torch.empty(H, dtype=torch.float32, device=dev)
        g_bc = torch.empty(H2, dtype=torch.float32, device=dev)
        none = (None,) * 5
        if FUSE_COEF_RECURRENCE and L >= 1 and Wc.stride(1) == 1:
            g2 = _f32c(g).contiguous()
            nws = int(lib.bg_d_coef_backward_workspace(H, ds, dl, L))
            n_g, n_w, n_t = (L + 1) * S * H2, 2 * H2 * H2, ((L * S + du) * H2 if cheb else 0)
            work = torch.empty(n_g + n_w + n_t + nws, dtype=torch.float32, device=dev)
            Gall, Wab = work[:n_g], work[n_g:n_g + n_w]
            shall = work[n_g + n_w:n_g + n_w + n_t] if cheb else None      # rebuilt there by one launch
            partial = work[n_g + n_w + n_t:]
            check(lib.bg_d_coef_backward_f32(ptr(g2), ptr(Xall), ptr(Wc), _ld(Wc), H, nf, nl, ML, ds, dl, L, basis,
                                             ptr(shall), ptr(Gall), ptr(Wab), ptr(partial), ptr(gW), ptr(g_bc), ptr(g_fe_w),
                                             ptr(g_fe_b), ptr(g_le_w), ptr(g_le_b), stream()), "bg_d_coef_backward_f32")
            return (gW, g_bc, g_fe_w, g_fe_b, g_le_w, g_le_b) + none
        Gall = torch.zeros(L + 1, S, H2, dtype=torch.float32, device=dev)
        Gall[L, :N].copy_(g)
        Wab = torch.cat([Wc[:, :H2], Wc[:, H2:2 * H2]], dim=0)       # [2*H2, H2]: G Wa + shiftup(G) Wb
        sht = torch.empty(N, H2, dtype=torch.float32, device=dev) if cheb else None
        for m in range(L - 1, -1, -1):
            nxt = Gall[m + 1]
            up = nxt[du:du + N]
            if cheb:
                cheb_shift_raw(nxt, sht, L + 1, L + 1, du * H2, True)
                up = sht
            gemm_raw([nxt[:N], up], Wab, False, False, residual=nxt[:N], out=Gall[m, :N])
        Gf = Gall[1:L + 1].reshape(L * S, H2)
        Xf = Xall.view((L + 1) * S, H2)
        if cheb and ShAll is None:                                    # the forward ran as one host call
            ShAll = _coef_shifts(Xall, L, du, H2)
        gemm_tn_raw(Gf, [Xf[du:du + L * S], ShAll[du:du + L * S] if cheb else Xf[:L * S]], out=gW[:, :2 * H2])
        check(lib.bg_d_coef_init_bwd_f32(ptr(Gall), S * H2, L, H, nf, nl, ML, ds, dl, ptr(g_fe_w), ptr(g_fe_b),
                                         ptr(g_le_w), ptr(g_le_b), gW.data_ptr() + 4 * 2 * H2, 2 * H2 + 4, ptr(g_bc),
                                         stream()), "bg_d_coef_init_bwd_f32")
        return (gW, g_bc, g_fe_w, g_fe_b, g_le_w, g_le_b) + none


class PolyCoefDW(torch.autograd.Function):
    """PolyCoefD with the block's composition (models.VoxelBlock.compose) inside the node: from messageMLP (Wm [2H, 4H+3],
    bm) and updateMLP (Wu [2H, 4H], bu) to the coefficient table in ONE host call (bg_d_table_forward_f32: prep, one
    product with a residual = [Wc | bc], the recurrence) and back (bg_d_table_backward_f32) -- instead of five stock
    launches and autograd nodes forward and seven backward around PolyCoefD.  Same arithmetic as compose() + PolyCoefD."""

    @staticmethod
    def forward(ctx, Wm, bm, Wu, bu, fe_w, fe_b, le_w, le_b, pe, L: int, ds: int, dl: int, basis=None):
        lib = _lib.load()
        _lib.require_device(Wm, "PolyCoefDW")
        H2 = Wm.shape[0]
        H, nf, nl, ML = H2 // 2, fe_w.shape[1], le_w.shape[1], pe.shape[0]
        if (H2 != 2 * H or Wm.shape[1] != 2 * H2 + 3 or tuple(Wu.shape) != (H2, 2 * H2) or fe_w.shape[0] != H
                or le_w.shape[0] != H or pe.shape[1] != H or ds < nf + ML + 5 or dl < nl or L < 1):
            raise _lib.BganError("PolyCoefDW: parameter shapes do not fit the collapsed discriminator")
        basis = D_BASIS if basis is None else basis
        du, P = ds + dl, 2 * H2 + 8
        N = (L + 1) * du
        S = N + du
        dev = Wm.device
        Wm2, Wu2 = _f32c(Wm), _f32c(Wu)
        args = [_f32c(t).contiguous() for t in (bm, bu, fe_w, fe_b, le_w, le_b, pe)]
        n_x, n_e, n_p = (L + 1) * S * H2, N * H2, H2 * P
        buf = torch.empty(n_x + n_e + 3 * n_p, dtype=torch.float32, device=dev)
        Xall = buf[:n_x].view(L + 1, S, H2)
        Efull = buf[n_x:n_x + n_e]
        wm_ext, r, wcb = (buf[n_x + n_e + i * n_p:n_x + n_e + (i + 1) * n_p].view(H2, P) for i in range(3))
        check(lib.bg_d_table_forward_f32(ptr(Wm2), _ld(Wm2), ptr(args[0]), ptr(Wu2), _ld(Wu2), ptr(args[1]), ptr(args[2]),
                                         ptr(args[3]), ptr(args[4]), ptr(args[5]), ptr(args[6]), H, nf, nl, ML, ds, dl, L,
                                         basis, ptr(wm_ext), ptr(r), ptr(wcb), ptr(Xall), ptr(Efull), stream()),
              "bg_d_table_forward_f32")
        ctx.dims = (L, du, ds, dl, H, nf, nl, ML, basis)
        ctx.save_for_backward(Xall, wcb, wm_ext, Wu2)
        return Xall[L, du:]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        Xall, wcb, wm_ext, Wu = ctx.saved_tensors
        L, du, ds, dl, H, nf, nl, ML, basis = ctx.dims
        cheb = basis == CHAIN_CHEB
        H2, P = 2 * H, 4 * H + 8
        N = (L + 1) * du
        S = N + du
        dev = g.device
        g_Wm = torch.empty(H2, 2 * H2 + 3, dtype=torch.float32, device=dev)
        g_Wu = torch.empty(H2, 2 * H2, dtype=torch.float32, device=dev)
        small = torch.empty(4 * H2 + H * (nf + nl), dtype=torch.float32, device=dev)
        g_bm, g_bu, g_fe_b, g_le_b = small[:H2], small[H2:2 * H2], small[2 * H2:2 * H2 + H], small[2 * H2 + H:3 * H2]
        g_fe_w = small[3 * H2:3 * H2 + H * nf].view(H, nf)
        g_le_w = small[3 * H2 + H * nf:3 * H2 + H * (nf + nl)].view(H, nl)
        g2 = _f32c(g).contiguous()
        nws = int(lib.bg_d_coef_backward_workspace(H, ds, dl, L))
        n_g, n_w, n_t, n_p = (L + 1) * S * H2, 2 * H2 * H2, ((L * S + du) * H2 if cheb else 0), H2 * P
        work = torch.empty(n_g + n_w + n_t + 2 * n_p + nws, dtype=torch.float32, device=dev)
        o = 0
        Gall = work[o:o + n_g]; o += n_g
        Wab = work[o:o + n_w]; o += n_w
        shall = work[o:o + n_t] if cheb else None; o += n_t
        gp = work[o:o + n_p]; o += n_p
        gwm = work[o:o + n_p]; o += n_p
        partial = work[o:]
        check(lib.bg_d_table_backward_f32(ptr(g2), ptr(Xall), ptr(wcb), ptr(wm_ext), ptr(Wu), _ld(Wu), H, nf, nl, ML, ds, dl,
                                          L, basis, ptr(shall), ptr(Gall), ptr(Wab), ptr(partial), ptr(gp), ptr(gwm),
                                          ptr(g_Wm), ptr(g_bm), ptr(g_Wu), ptr(g_bu), ptr(g_fe_w), ptr(g_fe_b), ptr(g_le_w),
                                          ptr(g_le_b), stream()), "bg_d_table_backward_f32")
        return (g_Wm, g_bm, g_Wu, g_bu, g_fe_w, g_fe_b, g_le_w, g_le_b) + (None,) * 5
