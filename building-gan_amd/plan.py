"""Per-batch segment plans, built once on the device and reused by every layer and every
generator / discriminator pass over the batch.

A plan replaces the COO bookkeeping the reference redoes inside every torch_scatter /
torch_geometric call (index_select + scatter pairs, ``torch.unique`` in avg_pool_x,
``batch.max().item()`` in global_max_pool: /root/reference/Model/models.py:24,65,71,158,166-167,
262-263).  Building it costs ONE host synchronisation per batch (reading back the validity flags).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import weakref
import os
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import _lib
from ._lib import check, ptr, stream

SCALE_NONE, SCALE_MUL, SCALE_DIV = 0, 1, 2
LONG_ROW = 32          # average row length above which the two-phase (chunked) kernels are used


@dataclass
class Csr:
    """CSR over ``n_rows`` segments whose slots point at rows of an ``n_cols``-row matrix."""
    rowptr: torch.Tensor                 # int32 [n_rows + 1]
    col: Optional[torch.Tensor]          # int32 [nnz]; None = slot k points at row k
    perm: Optional[torch.Tensor]         # int32 [nnz] original (COO) id of each slot; None = identity
    n_rows: int
    n_cols: int
    nnz: int
    count: Optional[torch.Tensor] = None       # float32 [n_rows] = max(len, 1)
    chunk_ptr: Optional[torch.Tensor] = None   # int32 [n_rows + 1]
    rows_per_chunk: int = 0
    n_chunks_ub: int = 0
    max_len: int = 0                           # longest segment (known for chunked CSRs after the plan's sync)
    seg_of: Optional[torch.Tensor] = None      # int32 [n_cols]: the segment of every row, where the segments partition the rows

    @property
    def long_rows(self) -> bool:
        return self.chunk_ptr is not None

    @property
    def row_per_workgroup(self) -> bool:
        """Many segments of moderate length (stories, program nodes) or a few short ones (the stories of
        a graph): one launch with a workgroup per segment instead of the two-phase chunk reduction."""
        return self.chunk_ptr is not None and ((self.n_rows >= 32 and 0 < self.max_len <= 4096) or 0 < self.max_len <= 256)


@dataclass
class VoxelBlocks:
    """Diagonal blocks of a square operator over the voxels: block b = rows [ptr[b], ptr[b+1])."""
    ptr: torch.Tensor          # int32 [n + 1]
    n: int
    max_rows: int
    max_nnz: int
    max_deg: int = 1 << 30     # longest row of the operator or of its transpose


class LinOp:
    """A sparse linear map  y = diag(rs) * P * diag(cs) * x  given by a Csr, with its transpose."""

    def __init__(self, csr: Csr, row_scale=None, row_mode=SCALE_NONE, col_scale=None, col_mode=SCALE_NONE):
        self.csr = csr
        self.row_scale, self.row_mode = row_scale, row_mode
        self.col_scale, self.col_mode = col_scale, col_mode
        self._T: Optional["LinOp"] = None      # the transpose, held by the pair's FIRST operator only
        self._back = None                      # in the transpose: (weak reference to the first operator, its Csr)

    @staticmethod
    def pair(fwd_csr: Csr, rev_csr: Csr, row_scale=None, row_mode=SCALE_NONE):
        """fwd = diag(rs) P (rows of fwd_csr); its transpose runs over rev_csr with a column scale.

        The two operators do not hold each other: a plan builds ~25 pairs per batch, and pairs in a reference cycle
        (with their Csr's and ~55 tensor views, i.e. the plan's device memory) were freed by the cyclic collector only
        -- 150 objects per train step for it to find, and the pauses of its full collections (tools/lab/mode_probe.py:
        two 50-70 ms stalls per 3,000 steps).  The first operator holds the transpose; the transpose finds the first
        one through a weak reference and, should that one be gone, makes it again from the Csr it keeps."""
        a = LinOp(fwd_csr, row_scale, row_mode)
        b = LinOp(rev_csr, None, SCALE_NONE, row_scale, row_mode)
        a._T = b
        b._back = (weakref.ref(a), fwd_csr)
        return a

    @property
    def T(self) -> Optional["LinOp"]:
        if self._T is not None or self._back is None:
            return self._T
        ref, fwd_csr = self._back
        a = ref()
        if a is None:                          # only the transpose was kept: the same first operator again
            a = LinOp(fwd_csr, self.col_scale, self.col_mode)
            a._T = self
            self._back = (weakref.ref(a), fwd_csr)
        return a


_IOTA = {}


def iota_i32(n: int, device) -> torch.Tensor:
    """0 .. n-1 as int32: a prefix of ONE array per device (the row pointers of one-slot-per-row operators are the
    same numbers for every batch; a torch.arange per plan was a launch per step).  The array is written once, and the
    device is synchronised then, so any stream may read it afterwards; it grows by doubling."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
    buf = _IOTA.get(key)
    if buf is None or buf.numel() < n:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return torch.arange(n, dtype=torch.int32, device=dev)      # (no synchronisation inside a capture: not kept)
        buf = torch.arange(max(2 * n, 1 << 16), dtype=torch.int32, device=dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        _IOTA[key] = buf
    return buf[:n]


def _i32(n, device):
    return torch.empty(n, dtype=torch.int32, device=device)


class _Builder:
    def __init__(self, device, max_builds=24):
        self.device = device
        self.lib = _lib.load()
        self.flags = torch.zeros(max_builds, 4, dtype=torch.int32, device=device)
        self.names = []
        self.chunked = []          # CSRs with chunk tables: their longest segment is read back in finish()

    def _slot(self, name):
        i = len(self.names)
        self.names.append(name)
        return self.flags[i]

    def csr(self, name, key, other, n_rows, n_cols, counts=False) -> Csr:
        """Stable sort of the COO list (key -> row, other -> col)."""
        E = int(key.numel())
        key = key.contiguous()
        other = other.contiguous()
        rowptr = _i32(n_rows + 1, self.device)
        col, perm, cursor = _i32(E, self.device), _i32(E, self.device), _i32(n_rows + 1, self.device)
        fl = self._slot(name)
        check(self.lib.bg_plan_csr(ptr(key), ptr(other), E, n_rows, ptr(rowptr), ptr(col), ptr(perm), ptr(cursor),
                                   ptr(fl), stream()), "bg_plan_csr")
        c = Csr(rowptr, col, perm, n_rows, n_cols, E)
        if counts:
            self.counts(c)
        self.maybe_chunks(c)
        return c

    def gather_csr(self, idx32: torch.Tensor, n_cols: int) -> Csr:
        """One slot per row: y[i] = x[idx[i]] (an index_select as a Csr)."""
        n = int(idx32.numel())
        return Csr(iota_i32(n + 1, self.device), idx32, None, n, n_cols, n)

    def segptr(self, name, seg_id, nseg) -> torch.Tensor:
        p = _i32(nseg + 1, self.device)
        fl = self._slot(name)
        check(self.lib.bg_plan_segptr(ptr(seg_id.contiguous()), int(seg_id.numel()), nseg, ptr(p), ptr(fl), stream()),
              "bg_plan_segptr")
        return p

    def counts(self, c: Csr) -> None:
        c.count = torch.empty(c.n_rows, dtype=torch.float32, device=self.device)
        check(self.lib.bg_plan_counts(ptr(c.rowptr), c.n_rows, ptr(c.count), stream()), "bg_plan_counts")

    def maybe_chunks(self, c: Csr, force: bool = False) -> None:
        if c.n_rows == 0 or (not force and c.nnz <= LONG_ROW * c.n_rows):
            return
        avg = c.nnz / max(c.n_rows, 1)
        R = 64 if avg <= 2048 else (256 if avg <= 32768 else 1024)
        c.chunk_ptr = _i32(c.n_rows + 1, self.device)
        c.rows_per_chunk = R
        c.n_chunks_ub = c.nnz // R + c.n_rows
        check(self.lib.bg_plan_chunks(ptr(c.rowptr), c.n_rows, R, ptr(c.chunk_ptr), stream()), "bg_plan_chunks")
        self.chunked.append(c)

    def finish(self, extra: torch.Tensor):
        """The one host synchronisation of a plan build: read the validity flags (and a few
        extra int32 scalars) back."""
        n = len(self.names)
        host = torch.cat([self.flags[:n].reshape(-1), extra.reshape(-1).to(torch.int32)]).cpu()
        fl = host[: 4 * n].view(n, 4)
        return {nm: tuple(int(x) for x in fl[i]) for i, nm in enumerate(self.names)}, host[4 * n:].tolist()


@dataclass
class GraphPlan:
    B: int = 0
    S: int = 0
    Np: int = 0
    Nv: int = 0
    Ep: int = 0
    Ev: int = 0
    Ec: int = 0
    device: Optional[torch.device] = None
    ops: dict = field(default_factory=dict)

    def __getattr__(self, name):
        ops = self.__dict__.get("ops", {})
        if name in ops:
            return ops[name]
        raise AttributeError(name)

    def begin_iteration(self):
        """Drop the data-derived tensors that the networks keep on the plan for the passes of ONE
        iteration (mean position differences, the discriminator's label-free operator powers), so
        that every training iteration recomputes them as it would for a fresh batch."""
        for k in [k for k in self.ops if k == "pdm4" or (isinstance(k, tuple) and k[0] == "d_static_powers")]:
            del self.ops[k]


def _max_plus_one(t: torch.Tensor) -> int:
    return int(t.max().item()) + 1 if t.numel() else 0


ONE_CALL_BUILD = os.environ.get("BG_PLAN_ONE_CALL", "1") != "0"     # bg_plan_build; "0": piece by piece
_I32 = {n: i for i, n in enumerate(_lib.PLAN_I32)}
# (ctypes array types made once: `c_int64 * n` in a function body builds a new class per call where n is not cached, and a
#  class is a reference cycle by itself)
_I64_I32, _I64x2, _I32x4 = C.c_int64 * len(_lib.PLAN_I32), C.c_int64 * 2, C.c_int32 * 4
_F32 = {n: i for i, n in enumerate(_lib.PLAN_F32)}
_I64_F32 = C.c_int64 * len(_lib.PLAN_F32)
_STAT = {n: i for i, n in enumerate(_lib.PLAN_STATS)}
_CSR_NAMES = ("program_by_target", "program_by_source", "class_slot", "slot_by_graph", "voxel_by_target", "voxel_by_source",
              "voxel_by_story", "voxel_by_graph", "story_by_graph", "cross_by_voxel", "cross_by_program")


def _chunk_rows(nnz, n_rows):
    avg = nnz / max(n_rows, 1)
    return 64 if avg <= 2048 else (256 if avg <= 32768 else 1024)


def _check_and_publish(graph, p, flags, min_cross):
    for name, fl in flags.items():
        if fl[0]:
            raise _lib.BganError(f"plan '{name}': {fl[0]} index values out of range")
    p.ops["flags"] = flags
    if flags["cross_by_voxel"][1] != 0:
        raise _lib.BganError("cross_edge_voxel_index_select must be non-decreasing (reference "
                             "GraphConstructor.py:81-88 builds it voxel-major)")
    if p.Nv > 0 and min_cross <= 0:
        raise _lib.BganError("every voxel needs at least one cross edge (reference util.py:24-25 indexes the "
                             "arg-max of each voxel segment)")
    if hasattr(graph, "__dict__"):
        graph.__dict__["_bg_plan"] = p
    return p


def _build_plan_one_call(graph, S, defer: bool = False):
    """The whole plan by bg_plan_build (include/bgan.h): two arenas, one host call, one read-back.  Returns None for
    batches it is not laid out for (empty pieces, operators with long rows where the piece-by-piece path would add
    chunk tables, derived operators above 2^22 slots): those take the piece-by-piece path below, which builds the
    same plan.  ``defer``: launch only -- the flags / statistics travel to pinned host memory asynchronously and a
    _PendingPlan is returned; its finish() reads them (prefetch_plan: the host never waits for the side stream)."""
    vf = graph.voxel_feature
    dev = vf.device
    pe, ve = graph.program_edge, graph.voxel_edge
    cep, cev = graph.cross_edge_program_index_select, graph.cross_edge_voxel_index_select
    vfc, vbi, pcc = graph.voxel_floor_cluster, graph.voxel_feature_batch, graph.program_class_cluster
    slot_graph, ratio = graph.program_target_ratio_batch, graph.program_target_ratio
    Np, Nv, B, n_slot = int(graph.program_class_feature.shape[0]), int(vf.shape[0]), int(graph.FAR.shape[0]), int(ratio.shape[0])
    Ep, Ev, Ec = int(pe.shape[1]), int(ve.shape[1]), int(cev.numel())
    if min(Np, Nv, Ev, Ec, B, S, n_slot) <= 0:
        return None
    # the piece-by-piece path adds chunk tables to any of these once its rows average more than LONG_ROW slots
    for nnz, rows in ((Ep, Np), (Np, n_slot), (n_slot, B), (Ev, Nv), (S, B), (Ec, Nv)):
        if nnz > LONG_ROW * rows:
            return None
    tensors = (pe, ve, cep, cev, vfc, vbi, pcc, slot_graph)
    if any(t.dtype != torch.int64 or not t.is_contiguous() for t in tensors) or ratio.dtype != torch.float32 \
            or not ratio.is_contiguous():
        return None
    lib = _lib.load()
    cp_R = _chunk_rows(Ec, Np) if Ec > LONG_ROW * Np else 0
    R = (_chunk_rows(Nv, S), _chunk_rows(Nv, B), _chunk_rows(S, B), cp_R)
    pin = _lib.PlanIn(Np, Nv, Ep, Ev, Ec, B, S, n_slot, pe.data_ptr(), pe.data_ptr() + 8 * Ep, ve.data_ptr(),
                      ve.data_ptr() + 8 * Ev, cep.data_ptr(), cev.data_ptr(), vfc.data_ptr(), vbi.data_ptr(), pcc.data_ptr(),
                      slot_graph.data_ptr(), ratio.data_ptr(), _I32x4(*R))
    oi, of, tot = _I64_I32(), _I64_F32(), _I64x2()
    rc = lib.bg_plan_layout(C.byref(pin), oi, of, tot)
    if rc == _lib.BG_ERR_SHAPE:
        return None
    check(rc, "bg_plan_layout")
    ai = torch.empty(tot[0], dtype=torch.int32, device=dev)
    af = torch.empty(tot[1], dtype=torch.float32, device=dev)
    check(lib.bg_plan_build(C.byref(pin), ai.data_ptr(), af.data_ptr(), stream()), "bg_plan_build")
    tail = ai[oi[_I32["FLAGS"]]: oi[_I32["STATS"]] + len(_STAT)]
    sizes = (Np, Nv, Ep, Ev, Ec, B, S, n_slot, cp_R, R)
    if defer:
        host_buf = torch.empty(tail.numel(), dtype=torch.int32, pin_memory=True)
        host_buf.copy_(tail, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        return _PendingPlan(dev, ai, af, list(oi), list(of), sizes, host_buf, done)
    return _finish_plan_one_call(graph, dev, ai, af, oi, of, sizes, tail.tolist())         # the one synchronisation


class _PendingPlan:
    """A plan whose kernels and read-back are in flight (plan.prefetch_plan)."""

    def __init__(self, dev, ai, af, oi, of, sizes, host_buf, done):      # (no reference to the batch: it holds this object)
        self.device, self.ai, self.af, self.oi, self.of = dev, ai, af, oi, of
        self.sizes, self.host_buf, self.done = sizes, host_buf, done

    def finish(self, graph) -> "GraphPlan":
        if not self.done.query():         # long over when the next iteration asks (the copy was queued a step ago);
            self.done.synchronize()       # a query is much cheaper than a synchronize that has nothing to wait for
        return _finish_plan_one_call(graph, self.device, self.ai, self.af, self.oi, self.of, self.sizes,
                                     self.host_buf.tolist())


def _finish_plan_one_call(graph, dev, ai, af, oi, of, sizes, host) -> "GraphPlan":
    Np, Nv, Ep, Ev, Ec, B, S, n_slot, cp_R, R = sizes
    flags = {nm: tuple(host[4 * i: 4 * i + 4]) for i, nm in enumerate(_CSR_NAMES)}
    st_off = oi[_I32["STATS"]] - oi[_I32["FLAGS"]]
    stat = {n: host[st_off + i] for n, i in _STAT.items()}

    def I(name, n):
        return ai.narrow(0, oi[_I32[name]], n)

    def F(name, n):
        return af.narrow(0, of[_F32[name]], n)

    def csr(pre, n_rows, n_cols, nnz, count=None):
        return Csr(I(pre + "_ROWPTR", n_rows + 1), I(pre + "_COL", nnz), I(pre + "_PERM", nnz), n_rows, n_cols, nnz,
                   count=None if count is None else F(count, n_rows))

    def chunked(c, pre, Rc, max_len):
        c.chunk_ptr, c.rows_per_chunk, c.n_chunks_ub, c.max_len = I(pre + "_CHUNK", c.n_rows + 1), Rc, c.nnz // Rc + c.n_rows, max_len

    def gather(name, n, n_cols):
        return Csr(I("ARANGE", n + 1), I(name, n), None, n, n_cols, n)

    p = GraphPlan(B=B, S=S, Np=Np, Nv=Nv, Ep=Ep, Ev=Ev, Ec=Ec, device=dev)
    o = p.ops
    o["_arenas"] = (ai, af)
    pf, pr = csr("PF", Np, Np, Ep, "PF_COUNT"), csr("PR", Np, Np, Ep)
    o["prog_mean"] = LinOp.pair(pf, pr, pf.count, SCALE_DIV)
    slot = csr("SLOT", n_slot, Np, Np, "SLOT_COUNT")
    slot_gather = gather("PCC32", Np, n_slot)
    o["slot_mean"] = LinOp.pair(slot, slot_gather, slot.count, SCALE_DIV)
    o["slot_sum"] = LinOp.pair(slot, slot_gather)
    o["slot_gather"] = o["slot_sum"].T
    sg = csr("SG", B, n_slot, n_slot)
    o["slot_graph_sum"] = LinOp.pair(sg, gather("SLOTGRAPH32", n_slot, B))
    o["slot_graph_gather"] = o["slot_graph_sum"].T
    vfw, vrv = csr("VFW", Nv, Nv, Ev, "VFW_COUNT"), csr("VRV", Nv, Nv, Ev)
    o["vox_mean"] = LinOp.pair(vfw, vrv, vfw.count, SCALE_DIV)
    st = csr("ST", S, Nv, Nv, "ST_COUNT")
    chunked(st, "ST", R[0], stat["MAXLEN_ST"])
    st_gather = gather("VFC32", Nv, S)
    o["story_mean"] = LinOp.pair(st, st_gather, st.count, SCALE_DIV)
    o["story_gather"] = LinOp.pair(st_gather, st)
    gr = csr("GR", B, Nv, Nv)
    chunked(gr, "GR", R[1], stat["MAXLEN_GR"])
    gr_gather = gather("VBI32", Nv, B)
    o["graph_gather"] = LinOp.pair(gr_gather, gr)
    st.seg_of, gr.seg_of = st_gather.col, gr_gather.col
    o["story_seg"], o["graph_seg"] = st, gr
    sgr = csr("SGR", B, S, S, "SGR_COUNT")
    sgr_seg = dataclasses.replace(sgr)
    chunked(sgr_seg, "SGR", R[2], stat["MAXLEN_SGR"])
    sgr_seg.seg_of = I("STORYGRAPH32", S)
    o["story_graph_seg"] = sgr_seg
    sgr_gather = gather("STORYGRAPH32", S, B)
    o["graph_of_story_mean"] = LinOp.pair(sgr, sgr_gather, sgr.count, SCALE_DIV)
    o["story_graph_gather"] = LinOp.pair(sgr_gather, sgr)
    o["level"] = I("LEVEL", Nv)
    cv, cp = csr("CV", Nv, Np, Ec), csr("CP", Np, Nv, Ec)
    cv.max_len = stat["MAXLEN_CV"]
    if cp_R:
        chunked(cp, "CP", cp_R, stat["MAXLEN_CP"])
    o["cross_v"], o["cross_p"] = cv, cp
    o["cross_v_sum"] = LinOp.pair(cv, cp)
    edge_by_voxel = Csr(cv.rowptr, cv.perm, None, Nv, Ec, Ec)
    edge_by_program = Csr(cp.rowptr, cp.perm, None, Np, Ec, Ec, chunk_ptr=cp.chunk_ptr, rows_per_chunk=cp.rows_per_chunk,
                          n_chunks_ub=cp.n_chunks_ub, max_len=cp.max_len)
    o["edge_voxel_gather"] = LinOp.pair(gather("CEV32", Ec, Nv), edge_by_voxel)
    o["edge_program_gather"] = LinOp.pair(gather("CEP32", Ec, Np), edge_by_program)
    nnz_ss, nnz_slot = stat["NNZ_SS"], stat["NNZ_SLOT"]
    if nnz_ss > 0:
        ss = Csr(I("SS_ROWPTR", S + 1), I("SS_COL", nnz_ss), None, S, S, nnz_ss, count=F("SS_COUNT", S))
        o["story_building_mean"] = LinOp(ss, ss.count, SCALE_DIV)
    o["prog_pair"] = LinOp(Csr(I("PP_ROWPTR", 2 * Np + 1), I("PP_COL", Ep + nnz_slot), None, 2 * Np, Np, Ep + nnz_slot),
                           F("PP_SCALE", 2 * Np), SCALE_MUL)
    if flags["voxel_by_graph"][2] == 0 and stat["N_LEAVING"] == 0:
        o["vox_blocks"] = VoxelBlocks(gr.rowptr, B, gr.max_len, stat["BLK_NNZ"], stat["VOX_MAX_DEG"])
    else:
        o["vox_blocks"] = None
    o["max_level"] = stat["MAX_LEVEL"]
    o["vox_all_has"], o["prog_all_has"] = stat["MIN_VDEG"] > 0, stat["MIN_PDEG"] > 0
    o["stories_all_nonempty"] = stat["MIN_STORY"] > 0
    o["vox_has"] = F("VOX_HAS", Nv).view(-1, 1)
    o["prog_has"] = F("PROG_HAS", Np).view(-1, 1)
    return _check_and_publish(graph, p, flags, stat["MIN_CROSS"])


_PREFETCH_STREAMS = {}


def prefetch_plan(graph, num_stories: Optional[int] = None) -> Optional[GraphPlan]:
    """Start the plan of a batch that a LATER iteration will use, beside the work already queued on the current
    stream: the launches go to a side stream, the flags / statistics go to pinned host memory by an asynchronous copy,
    and the host does NOT wait (what a loader's prefetch does for the reference's collate).  The first build_plan() on
    the batch reads the read-back (finished long before), builds the plan's views and makes the then-current stream
    wait for the side stream's event.  A plan depends on the batch's index vectors alone, never on parameters.
    Returns the plan when it already exists or had to be built synchronously, else None."""
    cached = graph.__dict__.get("_bg_plan") if hasattr(graph, "__dict__") else None
    if cached is not None:
        return cached
    if hasattr(graph, "__dict__") and "_bg_plan_pending" in graph.__dict__:
        return None
    dev = graph.voxel_feature.device
    if dev.type != "cuda":
        return build_plan(graph, num_stories)
    S1 = num_stories if num_stories is not None else getattr(graph, "num_stories", None)
    if not ONE_CALL_BUILD or S1 is None:
        # the piece-by-piece build starts with a read-back of its own (nothing would overlap) and allocates every plan
        # tensor one by one: it runs on the CURRENT stream, which owns those tensors for the iteration's kernels
        return build_plan(graph, num_stories)
    side = prefetch_stream(dev)
    main = torch.cuda.current_stream(dev)
    order_behind_producer(graph, side, main)
    with torch.cuda.stream(side):
        _lib.require_device(graph.voxel_feature, "prefetch_plan")
        pending = _build_plan_one_call(graph, int(S1), defer=True)
    if pending is None:
        # the one-call build declined the batch (long rows, an empty piece): as above, on the current stream
        return build_plan(graph, num_stories)
    for t in (pending.ai, pending.af):
        t.record_stream(main)             # allocated under the side stream, read by kernels of the main stream
    graph.__dict__["_bg_plan_pending"] = pending
    return None


def prefetch_stream(dev) -> "torch.cuda.Stream":
    side = _PREFETCH_STREAMS.get(dev)
    if side is None:
        side = _PREFETCH_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return side


_PLAN_INPUT_KEYS = ("program_edge", "voxel_edge", "cross_edge_program_index_select", "cross_edge_voxel_index_select",
                    "voxel_floor_cluster", "voxel_feature_batch", "program_class_cluster", "program_target_ratio_batch",
                    "program_target_ratio", "program_class_feature", "voxel_feature", "FAR")


def order_behind_producer(graph, side, main) -> None:
    """Order the side stream behind whatever produced the batch's tensors and tell the allocator that the side
    stream reads them.  A loader that copies from pinned memory with ``non_blocking=True`` (``VolumeDesignGraph.to``)
    leaves an event behind its last copy (``_bg_h2d_event``): the side stream waits for that event only, so the
    plan still runs beside the iteration queued before it.  Without the event the side stream waits for everything
    queued on the current stream so far (a batch built by device ops there is then complete as well)."""
    ev = graph.__dict__.get("_bg_h2d_event") if hasattr(graph, "__dict__") else None
    if ev is not None:
        side.wait_event(ev)
    else:
        side.wait_stream(main)
    for k in _PLAN_INPUT_KEYS:
        t = getattr(graph, k, None)
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(side)         # allocated under the main stream, read by the side stream's kernels


def build_plan(graph, num_stories: Optional[int] = None) -> GraphPlan:
    """Build (or fetch the cached) plan for a batch that exposes the row-a0 attributes."""
    cached = graph.__dict__.get("_bg_plan") if hasattr(graph, "__dict__") else None
    if cached is not None:
        ev = cached.ops.pop("_ready_event", None) if cached.ops else None
        if ev is not None:                # prefetched on a side stream: order this stream behind its launches
            torch.cuda.current_stream(cached.device).wait_event(ev)
        return cached
    pending = graph.__dict__.pop("_bg_plan_pending", None) if hasattr(graph, "__dict__") else None
    if pending is not None:               # launched by prefetch_plan: read the (long finished) flags, build the views
        torch.cuda.current_stream(pending.device).wait_event(pending.done)
        return pending.finish(graph)
    if ONE_CALL_BUILD:
        _lib.require_device(graph.voxel_feature, "build_plan")
        S1 = num_stories if num_stories is not None else getattr(graph, "num_stories", None)
        if S1 is None:
            S1 = _max_plus_one(graph.voxel_floor_cluster)
        made = _build_plan_one_call(graph, int(S1))
        if made is not None:
            return made
        num_stories = int(S1)
    vf = graph.voxel_feature
    _lib.require_device(vf, "build_plan")
    dev = vf.device
    b = _Builder(dev)
    pe, ve = graph.program_edge, graph.voxel_edge
    cep, cev = graph.cross_edge_program_index_select, graph.cross_edge_voxel_index_select
    vfc, vbi = graph.voxel_floor_cluster, graph.voxel_feature_batch
    pcc = graph.program_class_cluster
    Np, Nv = int(graph.program_class_feature.shape[0]), int(vf.shape[0])
    B = int(graph.FAR.shape[0])
    n_slot = int(graph.program_target_ratio.shape[0])
    S = num_stories if num_stories is not None else getattr(graph, "num_stories", None)
    if S is None:
        S = _max_plus_one(vfc)          # host sync, as the reference's own pooling does per call
    p = GraphPlan(B=B, S=S, Np=Np, Nv=Nv, Ep=int(pe.shape[1]), Ev=int(ve.shape[1]), Ec=int(cev.numel()), device=dev)
    o = p.ops

    # ---- program graph: mean over in-edges (target = edge[1]); class slots -----------------
    pf = b.csr("program_by_target", pe[1], pe[0], Np, Np, counts=True)
    pr = b.csr("program_by_source", pe[0], pe[1], Np, Np)
    o["prog_mean"] = LinOp.pair(pf, pr, pf.count, SCALE_DIV)
    ar_p = torch.arange(Np, dtype=torch.int64, device=dev)
    slot = b.csr("class_slot", pcc, ar_p, n_slot, Np, counts=True)
    slot_gather = b.gather_csr(pcc.to(torch.int32), n_slot)
    o["slot_mean"] = LinOp.pair(slot, slot_gather, slot.count, SCALE_DIV)      # avg_pool_x(size=6)
    o["slot_sum"] = LinOp.pair(slot, slot_gather)
    o["slot_gather"] = o["slot_sum"].T                                         # index_select(0, class_index)
    slot_graph = graph.program_target_ratio_batch
    ar_s = torch.arange(n_slot, dtype=torch.int64, device=dev)
    sg = b.csr("slot_by_graph", slot_graph, ar_s, B, n_slot)
    o["slot_graph_sum"] = LinOp.pair(sg, b.gather_csr(slot_graph.to(torch.int32), B))
    o["slot_graph_gather"] = o["slot_graph_sum"].T

    # ---- voxel graph ------------------------------------------------------------------------
    vfw = b.csr("voxel_by_target", ve[1], ve[0], Nv, Nv, counts=True)
    vrv = b.csr("voxel_by_source", ve[0], ve[1], Nv, Nv)
    o["vox_mean"] = LinOp.pair(vfw, vrv, vfw.count, SCALE_DIV)

    # stories / graphs: pooling segments over voxels (general CSR; contiguous when sorted)
    ar_v = torch.arange(Nv, dtype=torch.int64, device=dev)
    st = b.csr("voxel_by_story", vfc, ar_v, S, Nv, counts=True)
    b.maybe_chunks(st, force=True)
    st_gather = b.gather_csr(vfc.to(torch.int32), S)
    o["story_mean"] = LinOp.pair(st, st_gather, st.count, SCALE_DIV)
    o["story_gather"] = LinOp.pair(st_gather, st)
    gr = b.csr("voxel_by_graph", vbi, ar_v, B, Nv)
    b.maybe_chunks(gr, force=True)
    gr_gather = b.gather_csr(vbi.to(torch.int32), B)
    o["graph_gather"] = LinOp.pair(gr_gather, gr)
    st.seg_of, gr.seg_of = st_gather.col, gr_gather.col        # voxel -> story / graph (int32 [Nv])
    o["story_seg"], o["graph_seg"] = st, gr
    # graph id of a story = graph of (any) member voxel, as avg_pool_x's batch[perm]
    story_graph = torch.zeros(S, dtype=torch.int64, device=dev)
    story_graph.scatter_(0, vfc, vbi)
    ar_st = torch.arange(S, dtype=torch.int64, device=dev)
    sgr = b.csr("story_by_graph", story_graph, ar_st, B, S, counts=True)
    # pooling stories -> graphs (two-level global max): a copy with chunk tables (segment max needs them);
    # the mean over a graph's few stories keeps the one-launch short-row path
    sgr_seg = dataclasses.replace(sgr)
    b.maybe_chunks(sgr_seg, force=True)
    sgr_seg.seg_of = story_graph.to(torch.int32)               # story -> graph
    o["story_graph_seg"] = sgr_seg
    sgr_gather = b.gather_csr(story_graph.to(torch.int32), B)
    o["graph_of_story_mean"] = LinOp.pair(sgr, sgr_gather, sgr.count, SCALE_DIV)
    o["story_graph_gather"] = LinOp.pair(sgr_gather, sgr)

    # floor level of every voxel inside its own building
    first = _i32(max(B, 1), dev)
    level = _i32(Nv, dev)
    check(b.lib.bg_plan_floor_level(ptr(vfc.contiguous()), ptr(vbi.contiguous()), Nv, B, ptr(first), ptr(level),
                                    stream()), "bg_plan_floor_level")
    o["level"] = level
    lvl_max = level.max() if Nv > 0 else torch.zeros((), dtype=torch.int32, device=dev)

    # ---- cross edges ------------------------------------------------------------------------
    cv = b.csr("cross_by_voxel", cev, cep, Nv, Np)
    cp = b.csr("cross_by_program", cep, cev, Np, Nv)
    o["cross_v"], o["cross_p"] = cv, cp
    o["cross_v_sum"] = LinOp.pair(cv, cp)                     # rows = voxels, cols = program nodes
    # edge -> voxel / program gathers (y[e] = x[cev[e]]) and their transposes (segment sums over edges)
    edge_by_voxel = Csr(cv.rowptr, cv.perm, None, Nv, p.Ec, p.Ec)
    edge_by_program = Csr(cp.rowptr, cp.perm, None, Np, p.Ec, p.Ec, chunk_ptr=cp.chunk_ptr,
                          rows_per_chunk=cp.rows_per_chunk, n_chunks_ub=cp.n_chunks_ub)
    o["edge_voxel_gather"] = LinOp.pair(b.gather_csr(cev.to(torch.int32), Nv), edge_by_voxel)
    o["edge_program_gather"] = LinOp.pair(b.gather_csr(cep.to(torch.int32), Np), edge_by_program)

    def _min_len(c):
        return (c.rowptr[1:] - c.rowptr[:-1]).min() if c.n_rows > 0 else torch.ones((), dtype=torch.int32, device=dev)
    n_per_graph = (sgr.rowptr[1:] - sgr.rowptr[:-1]).to(torch.int64)            # stories of every graph
    n_per_slot = (slot.rowptr[1:] - slot.rowptr[:-1]).to(torch.int64)           # program nodes of every class slot
    measured = b.chunked + [cv]                       # CSRs whose longest segment the networks want to know
    extra = torch.stack([lvl_max.to(torch.int32), _min_len(vfw), _min_len(pf), _min_len(cv), _min_len(st)] +
                        [(c.rowptr[1:] - c.rowptr[:-1]).max() if c.n_rows > 0 else lvl_max.new_zeros(()).to(torch.int32)
                         for c in measured] + [(n_per_graph * n_per_graph).sum().to(torch.int32)] +
                        # diagonal blocks of the voxel operator (one per graph): edges that leave their graph, and
                        # the largest number of edges inside one graph
                        [(vbi[ve[0]] != vbi[ve[1]]).sum().to(torch.int32),
                         (lambda at: (at[1:] - at[:-1]).max() if B > 0 else lvl_max.new_zeros(()).to(torch.int32))(
                             vfw.rowptr[gr.rowptr.long()]),
                         torch.maximum((vfw.rowptr[1:] - vfw.rowptr[:-1]).max(), (vrv.rowptr[1:] - vrv.rowptr[:-1]).max())
                         if Nv > 0 else lvl_max.new_zeros(()).to(torch.int32),
                         (n_per_slot * n_per_slot).sum().to(torch.int32)])
    (flags, (max_level, min_vdeg, min_pdeg, min_cross, min_story, *max_lens, nnz_ss, n_leaving, blk_nnz, vox_max_deg,
             nnz_slot)) = b.finish(extra)
    # stories x stories: row s lists the stories of s's graph, scaled by 1 / their number -- the building
    # mean gathered back to the stories as ONE operator (its size is known only now, after the sync)
    nnz_ss = int(nnz_ss)
    if S > 0 and nnz_ss > 0:
        deg = n_per_graph[story_graph]                                            # [S]
        rp = torch.zeros(S + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(deg, 0)
        row_of = torch.repeat_interleave(ar_st, deg, output_size=nnz_ss)
        within = torch.arange(nnz_ss, dtype=torch.int64, device=dev) - rp[row_of]
        col_ss = sgr.col.to(torch.int64)[sgr.rowptr.to(torch.int64)[story_graph[row_of]] + within]
        ss = Csr(rp.to(torch.int32), col_ss.to(torch.int32), None, S, S, nnz_ss, count=deg.to(torch.float32))
        o["story_building_mean"] = LinOp(ss, ss.count, SCALE_DIV)
    # program graph, both aggregations of a ProgramBlock as ONE operator with 2 Np rows: row i = mean over the
    # in-edges of node i; row Np + i = target ratio of i's class slot x mean over the nodes of that slot
    # (avg_pool_x(size=6) * ratio, gathered back: reference models.py:71-72).  Scales are multipliers.
    nnz_slot = int(nnz_slot)
    if Np > 0:
        deg = n_per_slot[pcc]                                                     # [Np]
        rp = torch.zeros(Np + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(deg, 0)
        row_of = torch.repeat_interleave(ar_p, deg, output_size=nnz_slot)
        within = torch.arange(nnz_slot, dtype=torch.int64, device=dev) - rp[row_of]
        col_sl = slot.col.to(torch.int64)[slot.rowptr.to(torch.int64)[pcc[row_of]] + within]
        rowptr2 = torch.cat([pf.rowptr.to(torch.int64), p.Ep + rp[1:]]).to(torch.int32)
        col2 = torch.cat([pf.col, col_sl.to(torch.int32)])
        scale2 = torch.cat([1.0 / pf.count, graph.program_target_ratio.to(torch.float32).view(-1)[pcc] / slot.count[pcc]])
        o["prog_pair"] = LinOp(Csr(rowptr2, col2, None, 2 * Np, Np, p.Ep + nnz_slot), scale2.contiguous(), SCALE_MUL)
    for c, ml in zip(measured, max_lens):
        c.max_len = int(ml)
    edge_by_program.max_len = cp.max_len
    # voxel rows of a graph are contiguous and no voxel edge leaves its graph (PyG batching,
    # reference Data/VolumeDesignGraph.py:60-72): A and A^T are block diagonal, which lets a chain of
    # applications run as one launch (ops.power_chain_raw / horner_chain_raw)
    if B > 0 and Nv > 0 and flags["voxel_by_graph"][2] == 0 and n_leaving == 0:
        o["vox_blocks"] = VoxelBlocks(gr.rowptr, B, gr.max_len, int(blk_nnz), int(vox_max_deg))
    else:
        o["vox_blocks"] = None
    o["max_level"] = int(max_level)
    o["vox_all_has"], o["prog_all_has"] = min_vdeg > 0, min_pdeg > 0
    o["stories_all_nonempty"] = min_story > 0       # then max over a graph = max over its stories' maxima
    has = (vfw.rowptr[1:] > vfw.rowptr[:-1]).to(torch.float32).view(-1, 1)
    o["vox_has"] = has
    o["prog_has"] = (pf.rowptr[1:] > pf.rowptr[:-1]).to(torch.float32).view(-1, 1)
    return _check_and_publish(graph, p, flags, min_cross)
