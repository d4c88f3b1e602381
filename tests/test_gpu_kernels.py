"""Kernel-level parity of the HIP path (through the C-ABI) against the CPU oracle.

Bar: bit-exact for integer plans and arg-max indices; fp32 results within the stated tolerance
(rtol 1e-4 from BASELINE.json's north_star, tighter where the summation order is identical).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


def _batch(preset="small", seed=0, ragged=True):
    from building_gan_amd.synthetic import make_batch
    return make_batch(preset, seed=seed, ragged=ragged)


def _plan(batch, dev):
    from building_gan_amd.plan import build_plan
    g = batch.to(dev)
    return g, build_plan(g)


@pytest.mark.parametrize("preset,ragged", [("tiny", False), ("small", True), ("dataset", False)])
def test_plans_bit_exact(cuda, preset, ragged):
    from oracle import plan as oplan
    cpu = _batch(preset, 1, ragged)
    import copy
    g, p = _plan(copy.deepcopy(cpu), cuda)
    ve = cpu.voxel_edge.numpy()
    rp, col, perm = oplan.csr_by_key(ve[1], ve[0], p.Nv)
    c = p.vox_mean.csr
    assert np.array_equal(c.rowptr.cpu().numpy(), rp)
    assert np.array_equal(c.col.cpu().numpy(), col)
    assert np.array_equal(c.perm.cpu().numpy(), perm)
    rp, col, perm = oplan.csr_by_key(ve[0], ve[1], p.Nv)
    c = p.vox_mean.T.csr
    assert np.array_equal(c.rowptr.cpu().numpy(), rp) and np.array_equal(c.col.cpu().numpy(), col)
    assert np.array_equal(c.perm.cpu().numpy(), perm)
    deg = np.diff(oplan.csr_by_key(ve[1], ve[0], p.Nv)[0])
    assert np.array_equal(p.vox_mean.csr.count.cpu().numpy(), np.maximum(deg, 1).astype(np.float32))
    cep = cpu.cross_edge_program_index_select.numpy()
    cev = cpu.cross_edge_voxel_index_select.numpy()
    rp, col, perm = oplan.csr_by_key(cep, cev, p.Np)          # long rows: exercises the bitonic row sort
    c = p.cross_p
    assert np.array_equal(c.rowptr.cpu().numpy(), rp) and np.array_equal(c.col.cpu().numpy(), col)
    assert np.array_equal(c.perm.cpu().numpy(), perm)
    rp, col, perm = oplan.csr_by_key(cev, cep, p.Nv)
    assert np.array_equal(p.cross_v.rowptr.cpu().numpy(), rp) and np.array_equal(p.cross_v.col.cpu().numpy(), col)
    assert np.array_equal(p.story_seg.rowptr.cpu().numpy(), oplan.segment_ptr(cpu.voxel_floor_cluster.numpy(), p.S))
    assert np.array_equal(p.graph_seg.rowptr.cpu().numpy(), oplan.segment_ptr(cpu.voxel_feature_batch.numpy(), p.B))
    assert np.array_equal(p.level.cpu().numpy(),
                          oplan.floor_level(cpu.voxel_floor_cluster.numpy(), cpu.voxel_feature_batch.numpy()))
    for csr in (p.story_seg, p.graph_seg, p.cross_p):
        if csr.chunk_ptr is None:
            continue
        seg, st, en, cptr = oplan.chunk_table(csr.rowptr.cpu().numpy(), csr.rows_per_chunk)
        assert np.array_equal(csr.chunk_ptr.cpu().numpy(), cptr)
        assert csr.n_chunks_ub >= cptr[-1]


def _same(a, b, path, seen):
    """Field-by-field, bit-for-bit equality of two plans' pieces."""
    import dataclasses
    from building_gan_amd.plan import LinOp
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        assert isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor), path
        assert a.dtype == b.dtype and a.shape == b.shape, (path, a.dtype, b.dtype, a.shape, b.shape)
        assert torch.equal(a, b), path
        assert a.is_contiguous() and a.data_ptr() % 16 == 0 or a.numel() == 0, path
    elif dataclasses.is_dataclass(a):
        assert type(a) is type(b), path
        for f in dataclasses.fields(a):
            _same(getattr(a, f.name), getattr(b, f.name), f"{path}.{f.name}", seen)
    elif isinstance(a, LinOp):
        if id(a) in seen:
            return
        seen.add(id(a))
        assert isinstance(b, LinOp) and a.row_mode == b.row_mode and a.col_mode == b.col_mode, path
        for f in ("csr", "row_scale", "col_scale", "T"):
            _same(getattr(a, f), getattr(b, f), f"{path}.{f}", seen)
    elif isinstance(a, dict):
        assert set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            _same(a[k], b[k], f"{path}[{k}]", seen)
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("preset,ragged,seed", [("tiny", False, 0), ("small", True, 1), ("small", "jitter", 2),
                                                ("small", "jitter", 5), ("dataset", False, 3)])
def test_plan_in_one_host_call_equals_piece_by_piece_build(cuda, preset, ragged, seed, monkeypatch):
    """bg_plan_build (one host call, ~90 launches, no torch ops) makes the plan build_plan otherwise assembles from
    bg_plan_* calls and tensor operations: every index vector, table, count, flag and statistic, bit for bit."""
    from building_gan_amd import plan as P
    g = _batch(preset, seed, ragged).to(cuda)
    monkeypatch.setattr(P, "ONE_CALL_BUILD", True)
    fast = P.build_plan(g)
    assert "_arenas" in fast.ops, "the one-call path declined this batch"
    g.__dict__.pop("_bg_plan")
    monkeypatch.setattr(P, "ONE_CALL_BUILD", False)
    slow = P.build_plan(g)
    assert "_arenas" not in slow.ops
    a = {k: v for k, v in fast.ops.items() if k != "_arenas"}
    _same(a, dict(slow.ops), "plan", set())
    for f in ("B", "S", "Np", "Nv", "Ep", "Ev", "Ec"):
        assert getattr(fast, f) == getattr(slow, f)


def test_plan_in_one_host_call_reports_bad_indices(cuda, monkeypatch):
    """Out-of-range ids in any index vector end in the plan's error, not in a stray access."""
    from building_gan_amd import _lib, plan as P
    monkeypatch.setattr(P, "ONE_CALL_BUILD", True)
    for field, value in (("voxel_feature_batch", 10 ** 6), ("program_class_cluster", -3), ("voxel_floor_cluster", 10 ** 7),
                         ("cross_edge_program_index_select", 10 ** 8), ("program_target_ratio_batch", 99)):
        g = _batch("small", 4, True).to(cuda)
        S = int(g.voxel_floor_cluster.max()) + 1
        # the plan's arenas are torch.empty: make what the allocator hands back hold wild values, so that a slot the
        # build leaves unwritten (edges with a bad key are never placed) would show if anything read it as an index
        junk = torch.full((1 << 22,), 0x7f7f7f7f, dtype=torch.int32, device=cuda)
        del junk
        t = getattr(g, field).clone()
        t[t.numel() // 2] = value
        setattr(g, field, t)
        with pytest.raises(_lib.BganError):
            P.build_plan(g, num_stories=S)
    torch.cuda.synchronize()


def test_prefetched_plan_equals_direct_build_and_orders_the_consumer(cuda):
    """plan.prefetch_plan launches on a side stream and does NOT wait for the read-back; the first build_plan() on the
    batch finishes it: that plan must equal the directly built one bit for bit, and the current stream must be ordered
    behind the side stream.  A batch without a story count takes the synchronous prefetch (event on the plan)."""
    from building_gan_amd import ops, plan as P
    g = _batch("dataset", 9, "jitter").to(cuda)
    direct = P.build_plan(g)
    g.__dict__.pop("_bg_plan")
    x = torch.rand(direct.Nv, 128, device=cuda)
    want = ops.spmm_raw(direct.vox_mean, x)
    # keep the main stream busy so that the side stream really runs beside it
    for _ in range(20):
        ops.spmm_raw(direct.vox_mean, x)
    assert getattr(g, "num_stories", None) is not None
    assert P.prefetch_plan(g) is None and "_bg_plan_pending" in g.__dict__ and "_bg_plan" not in g.__dict__
    assert P.prefetch_plan(g) is None                              # idempotent while pending
    got_plan = P.build_plan(g)
    assert "_bg_plan_pending" not in g.__dict__ and g.__dict__["_bg_plan"] is got_plan
    got = ops.spmm_raw(got_plan.vox_mean, x)
    assert torch.equal(got, want)
    a = {k: v for k, v in got_plan.ops.items() if k != "_arenas"}
    b = {k: v for k, v in direct.ops.items() if k != "_arenas"}
    _same(a, b, "plan", set())
    # moving the batch drops a pending plan like a finished one
    P.prefetch_plan(_drop(g))
    g.to(cuda)                                                     # already there: nothing moves, the pending plan stays
    assert "_bg_plan_pending" in g.__dict__
    g.to("cpu")
    g.to(cuda)
    assert "_bg_plan_pending" not in g.__dict__ and "_bg_plan" not in g.__dict__
    kept = P.build_plan(g)
    assert g.to(cuda) is g and g.__dict__.get("_bg_plan") is kept  # a resident batch keeps its finished plan as well
    # no story count on the batch: the synchronous form, built on the CURRENT stream (it starts with a read-back, so
    # nothing would overlap, and its tensors must belong to the stream whose kernels read them)
    g.__dict__.pop("num_stories", None)
    pre = P.prefetch_plan(g)
    assert pre is not None and "_ready_event" not in pre.ops
    assert P.build_plan(g) is pre
    _same({k: v for k, v in pre.ops.items() if k != "_arenas"}, b, "plan", set())
    # bad indices surface when the pending plan is finished
    bad = _batch("small", 3, True).to(cuda)
    bad.cross_edge_voxel_index_select = bad.cross_edge_voxel_index_select.flip(0).contiguous()
    if getattr(bad, "num_stories", None) is not None:
        assert P.prefetch_plan(bad) is None
        with pytest.raises(Exception, match="non-decreasing"):
            P.build_plan(bad)


def test_prefetched_plan_waits_for_a_non_blocking_copy_from_pinned_memory(cuda):
    """A loader hands over pinned host tensors and copies them with non_blocking=True: the host runs ahead of the copy
    engine.  prefetch_plan's side stream must order itself behind the batch's copies (the event VolumeDesignGraph.to
    leaves behind them) -- otherwise k_hist_b reads index vectors that have not arrived."""
    import copy
    from building_gan_amd import ops, plan as P
    cpu = _batch("dataset", 11, "jitter")
    direct = P.build_plan(copy.deepcopy(cpu).to(cuda))
    want = {k: v for k, v in direct.ops.items() if k != "_arenas"}
    x = torch.rand(direct.Nv, 128, device=cuda)
    filler = torch.empty(1 << 26, dtype=torch.uint8).pin_memory()
    sink = torch.empty(1 << 26, dtype=torch.uint8, device=cuda)
    for rep in range(4):
        b = copy.deepcopy(cpu).pin_memory()
        assert b.voxel_edge.is_pinned()
        # work queued in front: a 64 MiB upload on the same stream (the batch's copies wait behind it) and kernels
        sink.copy_(filler, non_blocking=True)
        for _ in range(8):
            ops.spmm_raw(direct.vox_mean, x)
        b.to(cuda, non_blocking=True)
        assert "_bg_h2d_event" in b.__dict__
        assert P.prefetch_plan(b) is None and "_bg_plan_pending" in b.__dict__
        got = P.build_plan(b)
        _same({k: v for k, v in got.ops.items() if k != "_arenas"}, want, "plan[%d]" % rep, set())
    # a blocking move leaves the event too (the plan then runs beside whatever the current stream still holds) ..
    b = copy.deepcopy(cpu).to(cuda)
    assert "_bg_h2d_event" in b.__dict__
    for _ in range(8):
        ops.spmm_raw(direct.vox_mean, x)
    assert P.prefetch_plan(b) is None
    _same({k: v for k, v in P.build_plan(b).ops.items() if k != "_arenas"}, want, "plan[blocking]", set())
    # .. and a batch without one (assembled by device ops on the current stream) makes the side stream wait for that stream
    b = copy.deepcopy(cpu).to(cuda)
    b.__dict__.pop("_bg_h2d_event")
    b.voxel_edge = b.voxel_edge.clone()
    assert P.prefetch_plan(b) is None
    _same({k: v for k, v in P.build_plan(b).ops.items() if k != "_arenas"}, want, "plan[device ops]", set())
    torch.cuda.synchronize()


def test_prefetch_of_the_piece_by_piece_plan_stays_on_the_current_stream(cuda, monkeypatch):
    """With the one-call build switched off (or declined) prefetch_plan builds synchronously on the current stream:
    every plan tensor then belongs to the stream whose kernels read it, so dropping the plan while those kernels are
    still queued cannot hand its blocks to a later prefetch on another stream."""
    from building_gan_amd import ops, plan as P
    monkeypatch.setattr(P, "ONE_CALL_BUILD", False)
    g = _batch("small", 6, "jitter").to(cuda)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # "the current stream" need not be the default one
        p = P.prefetch_plan(g)
        assert p is not None and "_arenas" not in p.ops and "_ready_event" not in p.ops
        assert "_bg_plan_pending" not in g.__dict__ and g.__dict__["_bg_plan"] is p
        x = torch.rand(p.Nv, 128, device=cuda)
        first = ops.spmm_raw(p.vox_mean, x)
        # drop the plan with kernels that read it still queued, build another batch's plan, run the first result again
        rowptr, col = p.vox_mean.csr.rowptr.clone(), p.vox_mean.csr.col.clone()
        for _ in range(50):
            ops.spmm_raw(p.vox_mean, x)
        last = ops.spmm_raw(p.vox_mean, x)
        g.__dict__.pop("_bg_plan")
        del p
        other = _batch("small", 7, "jitter").to(cuda)
        P.prefetch_plan(other)
        assert torch.equal(first, last)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert int(rowptr[-1]) == col.numel()


def _drop(g):
    g.__dict__.pop("_bg_plan", None)
    g.__dict__.pop("_bg_plan_pending", None)
    return g


def test_plan_unsorted_random_keys(cuda):
    """Random (unsorted, skewed) keys incl. empty rows and one very long row (> LDS sort size)."""
    from building_gan_amd.plan import _Builder
    from oracle import plan as oplan
    rng = np.random.default_rng(3)
    N, E = 1000, 60000
    key = rng.integers(0, N, size=E)
    key[rng.random(E) < 0.15] = 7            # ~9000 edges in one row -> global-memory bitonic path
    key[key == 11] = 12                      # an empty row
    other = rng.integers(0, 500, size=E)
    b = _Builder(cuda)
    c = b.csr("t", torch.from_numpy(key).to(cuda), torch.from_numpy(other).to(cuda), N, 500)
    flags, _ = b.finish(torch.zeros(1, dtype=torch.int32, device=cuda))
    rp, col, perm = oplan.csr_by_key(key, other, N)
    assert flags["t"][0] == 0
    assert np.array_equal(c.rowptr.cpu().numpy(), rp)
    assert np.array_equal(c.perm.cpu().numpy(), perm)
    assert np.array_equal(c.col.cpu().numpy(), col)


@pytest.mark.parametrize("H", [128, 256, 8, 6, 1])
def test_gather_mean_matches_scatter_mean(cuda, H):
    from building_gan_amd import ops
    from oracle.scatter import scatter_mean
    import copy
    cpu = _batch("small", 2, True)
    g, p = _plan(copy.deepcopy(cpu), cuda)
    torch.manual_seed(0)
    x = torch.rand(p.Nv, H) * 2 - 1
    ref = scatter_mean(x.index_select(0, cpu.voxel_edge[0]), cpu.voxel_edge[1], p.Nv)
    out = ops.spmm_raw(p.vox_mean, x.to(cuda)).cpu()
    # same summation order + IEEE division: expected to agree to the last bit
    assert torch.equal(out, ref)
    # transposed operator against autograd of the oracle
    xg = x.clone().requires_grad_(True)
    go = torch.rand(p.Nv, H) * 2 - 1
    scatter_mean(xg.index_select(0, cpu.voxel_edge[0]), cpu.voxel_edge[1], p.Nv).backward(go)
    outT = ops.spmm_raw(p.vox_mean.T, go.to(cuda)).cpu()
    torch.testing.assert_close(outT, xg.grad, rtol=RTOL, atol=ATOL)


def test_pooling_and_segmax(cuda):
    from building_gan_amd import ops
    from oracle.scatter import avg_pool_x, global_max_pool, scatter_max
    import copy
    cpu = _batch("small", 4, True)
    g, p = _plan(copy.deepcopy(cpu), cuda)
    torch.manual_seed(1)
    H = 256
    x = torch.rand(p.Nv, H) * 2 - 1
    x[3] = x[5]                                   # ties inside a segment -> first index must win
    xd = x.to(cuda)
    story_ref, story_batch = avg_pool_x(cpu.voxel_floor_cluster, x, cpu.voxel_feature_batch)
    story = ops.spmm_raw(p.story_mean, xd)
    torch.testing.assert_close(story.cpu(), story_ref, rtol=RTOL, atol=ATOL)
    bld_ref = avg_pool_x(story_batch, story_ref, torch.zeros(story_ref.shape[0], dtype=torch.long))[0]
    bld = ops.spmm_raw(p.graph_of_story_mean, story)
    torch.testing.assert_close(bld.cpu(), bld_ref, rtol=RTOL, atol=ATOL)
    sv = ops.spmm_raw(p.story_gather, story).cpu()
    torch.testing.assert_close(sv, story_ref.index_select(0, cpu.voxel_floor_cluster), rtol=RTOL, atol=ATOL)
    for csr, idx, n in ((p.story_seg, cpu.voxel_floor_cluster, p.S), (p.graph_seg, cpu.voxel_feature_batch, p.B)):
        val, arg = ops.segmax_raw(csr, xd)
        rv, ra = scatter_max(x, idx, n)
        assert torch.equal(val.cpu(), rv)                      # max is exact
        assert torch.equal(arg.cpu().long(), ra)               # first index on ties, bit-exact
        assert torch.equal(global_max_pool(x, idx), val.cpu())


def test_segmax_empty_segment(cuda):
    from building_gan_amd import ops
    from building_gan_amd.plan import _Builder
    from oracle.scatter import scatter_max
    seg = torch.tensor([0, 0, 2, 2, 2, 4], dtype=torch.long)       # segments 1 and 3 are empty
    x = torch.tensor([[1., -2.], [1., 5.], [-3., -3.], [-1., -4.], [-1., -3.], [0., 0.]])
    b = _Builder(cuda)
    c = b.csr("s", seg.to(cuda), torch.arange(6).to(cuda), 5, 6)
    b.maybe_chunks(c, force=True)
    val, arg = ops.segmax_raw(c, x.to(cuda))
    rv, ra = scatter_max(x, seg, 5)
    assert torch.equal(val.cpu(), rv)
    ra = torch.where(ra == 6, torch.full_like(ra, -1), ra)         # oracle sentinel E -> library sentinel -1
    assert torch.equal(arg.cpu().long(), ra)


@pytest.mark.parametrize("shape", [(300, 128, [128, 128, 4]), (1000, 256, [256, 256, 4]), (77, 2, [64]),
                                   (513, 128, [39]), (2000, 128, [128, 128, 128, 128]), (130, 64, [128]),
                                   (20000, 128, [128, 128, 4]),
                                   # narrow last segments: unaligned widths, a skinny-kernel shape, the 128x128-tile shape
                                   (400, 96, [64, 36, 3]), (500, 128, [128, 2]), (90, 64, [128, 128, 4]),
                                   (70000, 256, [256, 256, 4]), (3000, 128, [128, 1]),
                                   (70000, 128, [128, 128, 4])])      # narrow output, many rows: 128x64 tiles
def test_gemm_linear_cat(cuda, shape):
    from building_gan_amd import ops
    M, N, Ks = shape
    torch.manual_seed(M)
    xs = [torch.rand(M, k) * 2 - 1 for k in Ks]
    W = torch.rand(N, sum(Ks)) * 2 - 1
    b = torch.rand(N) * 2 - 1
    rs = (torch.rand(M) > 0.2).float()
    ref = (torch.cat(xs, 1).double() @ W.double().t() + b.double()) * rs.double().view(-1, 1)
    ref = torch.nn.functional.leaky_relu(ref, 0.01)
    out = ops.gemm_raw([x.to(cuda) for x in xs], W.to(cuda), False, True, bias=b.to(cuda), row_scale=rs.to(cuda),
                       act=ops.ACT_LEAKY, slope=0.01)
    scale = torch.cat(xs, 1).abs().double() @ W.abs().double().t()
    err = (out.cpu().double() - ref).abs()
    assert float((err / (scale + 1.0)).max()) < 2e-6          # fp32 fma chain vs fp64


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_layouts(cuda, ta, tb):
    from building_gan_amd import ops
    torch.manual_seed(5)
    M, N, K = 333, 200, 1028
    A = torch.rand(K, M) * 2 - 1 if ta else torch.rand(M, K) * 2 - 1
    B = torch.rand(N, K) * 2 - 1 if tb else torch.rand(K, N) * 2 - 1
    ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    out = ops.MatMul.apply(A.to(cuda), B.to(cuda), ta, tb).cpu().double()
    assert float((out - ref).abs().max()) < 1e-3 * 1e-1 * K ** 0.5
    # asymmetric identity check (catches transposed C writes)
    eye = torch.eye(64)
    Bm = torch.arange(64 * 48, dtype=torch.float32).view(64, 48)
    out = ops.gemm_raw([eye.to(cuda)], Bm.to(cuda), False, False).cpu()
    assert torch.equal(out, Bm)


@pytest.mark.parametrize("M,N,K", [(11520, 128, 256), (2050, 128, 64), (300, 70, 36), (64, 64, 8)])
def test_gemm_accumulator_start(cuda, M, N, K):
    """bg_gemm_f32's acc_init: the product starts from a given [M, N] matrix (before bias / activation /
    residual), against the fp64 statement; ragged edge tiles included."""
    from building_gan_amd import ops
    torch.manual_seed(62)
    A = torch.rand(M, K) * 2 - 1
    W = torch.rand(N, K) * 2 - 1
    Q = torch.rand(M, N) * 2 - 1
    b = torch.rand(N) * 2 - 1
    R = torch.rand(M, N) * 2 - 1
    y = torch.full((M, N), -7.0, device=cuda)
    out_d = ops.gemm_raw([A.to(cuda)], W.to(cuda), False, True, bias=b.to(cuda), act=ops.ACT_LEAKY, slope=0.01,
                         residual=R.to(cuda), acc_init=Q.to(cuda), pre_residual_out=y)
    out = out_d.cpu().double()
    z = Q.double() + A.double() @ W.double().t() + b.double()
    act = torch.where(z > 0, z, 0.01 * z)
    ref = act + R.double()
    assert float((out - ref).abs().max()) < 2e-5 * K ** 0.5
    # the second output holds the value before the residual: out == y + R exactly (one fp32 addition)
    assert float((y.cpu().double() - act).abs().max()) < 2e-5 * K ** 0.5
    assert torch.equal(out_d, y + R.to(cuda))


def test_gemm_tn_splitk_and_colsum(cuda):
    from building_gan_amd import ops
    torch.manual_seed(6)
    R, M, N = 11520, 256, 516
    A = torch.rand(R, M) * 2 - 1
    B = torch.rand(R, N) * 2 - 1
    out = ops.gemm_tn_raw(A.to(cuda), B.to(cuda)).cpu().double()
    ref = A.double().t() @ B.double()
    assert float((out - ref).abs().max()) < 2e-3
    cs = ops.colsum_raw(A.to(cuda)).cpu().double()
    assert float((cs - A.double().sum(0)).abs().max()) < 2e-3


@pytest.mark.parametrize("R,M,N", [(11520, 128, 260), (90, 128, 256), (11520, 132, 128), (64, 8, 3)])
def test_gemm_tn_and_colsum_accumulate_into_a_buffer(cuda, R, M, N):
    """accumulate != 0: the product / column sum is ADDED to the buffer (exactly buffer + what the plain call
    writes, one fp32 addition), through every reduce kernel the split-K launcher picks."""
    from building_gan_amd import ops
    torch.manual_seed(61)
    A = (torch.rand(R, M) * 2 - 1).to(cuda)
    B = (torch.rand(R, N) * 2 - 1).to(cuda)
    base = (torch.rand(M, N) * 2 - 1).to(cuda)
    plain = ops.gemm_tn_raw(A, B)
    buf = base.clone()
    assert ops.gemm_tn_raw(A, B, out=buf, accumulate=True) is buf
    assert torch.equal(buf, base + plain)
    ops.gemm_tn_raw(A, B, out=buf, accumulate=False)
    assert torch.equal(buf, plain)
    vb = (torch.rand(M) * 2 - 1).to(cuda)
    vplain = ops.colsum_raw(A)
    v = vb.clone()
    ops.colsum_raw(A, out=v, accumulate=True)
    assert torch.equal(v, vb + vplain)
    ops.colsum_raw(A, out=v)
    assert torch.equal(v, vplain)
    with pytest.raises(Exception):
        ops.gemm_tn_raw(A, B, accumulate=True)


def test_segment_gumbel_softmax_and_argmax(cuda):
    from building_gan_amd import ops
    from oracle.pointer import segment_gumbel_softmax, find_max_out_program_index
    import copy
    cpu = _batch("dataset", 7, False)
    g, p = _plan(copy.deepcopy(cpu), cuda)
    cev, cep = cpu.cross_edge_voxel_index_select, cpu.cross_edge_program_index_select
    torch.manual_seed(7)
    att = (torch.rand(p.Ec) * 2 - 1).requires_grad_(True)
    noise = torch.empty(p.Ec).exponential_()
    soft_ref, hard_ref = segment_gumbel_softmax(att, cev, exp_noise=noise)
    attd = att.detach().to(cuda).requires_grad_(True)
    soft, hard, arg = ops.SegGumbelSoftmax.apply(attd, noise.to(cuda), p, -1.0)
    torch.testing.assert_close(soft.cpu(), soft_ref.detach(), rtol=RTOL, atol=1e-7)
    torch.testing.assert_close(hard.cpu(), hard_ref.detach(), rtol=RTOL, atol=1e-6)
    # the pointer itself: bit-exact
    from oracle.scatter import scatter_max
    arg_ref = scatter_max(soft_ref.detach(), cev, p.Nv)[1]
    assert torch.equal(arg.cpu().long(), arg_ref)
    # backward of (soft, hard) with random cotangents
    g1, g2 = torch.rand(p.Ec), torch.rand(p.Ec)
    (soft_ref * g1 + hard_ref * g2).sum().backward()
    (soft * g1.to(cuda) + hard * g2.to(cuda)).sum().backward()
    torch.testing.assert_close(attd.grad.cpu(), att.grad, rtol=1e-3, atol=1e-6)
    # arg-max program index with masked voxels (all-zero segments tie -> first edge)
    mask = (torch.rand(p.Nv, 1) > 0.5).float()
    logit = hard_ref.detach().view(-1, 1) * mask.index_select(0, cev)
    idx_ref = find_max_out_program_index(logit, cev, cep, p.Nv).view(-1)
    a2 = ops.seg_argmax(logit.to(cuda), p)
    idx = g.cross_edge_program_index_select[a2.long()]
    assert torch.equal(idx.cpu(), idx_ref)


def test_cross_logits_forward_backward(cuda):
    from building_gan_amd import ops
    import copy
    cpu = _batch("small", 8, True)
    g, p = _plan(copy.deepcopy(cpu), cuda)
    cev, cep = cpu.cross_edge_voxel_index_select, cpu.cross_edge_program_index_select
    torch.manual_seed(8)
    H = 128
    a = (torch.rand(p.Nv, H) * 2 - 1).requires_grad_(True)
    b = (torch.rand(p.Np, H) * 2 - 1).requires_grad_(True)
    th = (torch.rand(1, H) * 2 - 1).requires_grad_(True)
    ref = (th.view(1, -1) * torch.tanh(a.index_select(0, cev) + b.index_select(0, cep))).sum(-1)
    go = torch.rand(p.Ec) * 2 - 1
    ref.backward(go)
    ad, bd, thd = (t.detach().to(cuda).requires_grad_(True) for t in (a, b, th))
    out = ops.CrossLogits.apply(ad, bd, thd, p)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=RTOL, atol=1e-5)
    out.backward(go.to(cuda))
    torch.testing.assert_close(ad.grad.cpu(), a.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(bd.grad.cpu(), b.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(thd.grad.cpu(), th.grad, rtol=1e-3, atol=1e-4)


def test_attention_tanh_over_the_whole_line(cuda):
    """The kernels' tanh ((e - 1) / (e + 1) on the hardware exp2 / reciprocal, common.h bg_tanh) against fp64 tanh:
    a one-hot theta makes every cross-edge logit ONE tanh value, the arguments sweep 1e-7 .. 100 of both signs, +-0, +-1e4.
    Stated bound: absolute error <= 1.5e-7 everywhere (the oracle's fp32 tanh is itself within 6e-8)."""
    from building_gan_amd import ops
    import copy
    cpu = _batch("small", 8, True)
    g, p = _plan(copy.deepcopy(cpu), cuda)
    cev = cpu.cross_edge_voxel_index_select
    H = 128
    mag = torch.logspace(-7, 2, p.Nv, dtype=torch.float64)
    xs = (mag * torch.where(torch.arange(p.Nv) % 2 == 0, 1.0, -1.0)).float()
    xs[:4] = torch.tensor([0.0, -0.0, 1e4, -1e4])
    a = torch.zeros(p.Nv, H); a[:, 0] = xs
    th = torch.zeros(1, H); th[0, 0] = 1.0
    out = ops.CrossLogits.apply(a.to(cuda), torch.zeros(p.Np, H, device=cuda), th.to(cuda), p).cpu()
    want = torch.tanh(xs.double())[cev]
    err = (out.double() - want).abs()
    assert float(err.max()) <= 1.5e-7, float(err.max())
    big = xs.abs()[cev] >= 0.3
    assert float((err[big] / want[big].abs()).max()) <= 2.5e-7            # ~2 ulp where tanh is of order one
    assert bool((out.abs() <= 1).all())                                   # (so 1 - t^2 >= 0 in the backward)
    sat = out[xs[cev].abs() == 1e4]
    assert sat.numel() > 0 and bool((sat.abs() >= 1 - 1.2e-7).all()) and torch.equal(sat.sign(), xs[cev][xs[cev].abs() == 1e4].sign())


def test_weighted_sum_and_sddmm(cuda):
    from building_gan_amd import ops
    from oracle.scatter import scatter_sum
    import copy
    cpu = _batch("small", 9, True)
    g, p = _plan(copy.deepcopy(cpu), cuda)
    cev, cep = cpu.cross_edge_voxel_index_select, cpu.cross_edge_program_index_select
    torch.manual_seed(9)
    for H in (128, 6):
        x = (torch.rand(p.Np, H) * 2 - 1).requires_grad_(True)
        w = torch.rand(p.Ec).requires_grad_(True)
        ref = scatter_sum(w.view(-1, 1) * x[cep], cev, p.Nv)
        go = torch.rand(p.Nv, H) * 2 - 1
        ref.backward(go)
        xd, wd = x.detach().to(cuda).requires_grad_(True), w.detach().to(cuda).requires_grad_(True)
        out = ops.SpMMW.apply(xd, wd, p.cross_v_sum)
        torch.testing.assert_close(out.cpu(), ref.detach(), rtol=RTOL, atol=1e-5)
        out.backward(go.to(cuda))
        torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(wd.grad.cpu(), w.grad, rtol=1e-3, atol=1e-5)


def test_adam_matches_torch(cuda):
    from building_gan_amd import ops
    torch.manual_seed(10)
    n = 10007
    p0 = torch.rand(n) * 2 - 1
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-4, betas=(0.5, 0.999))
    p, m, v = p0.to(cuda), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    for step in range(1, 4):
        gr = torch.rand(n) * 2 - 1
        ref.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr.to(cuda), m, v, 1e-4, 0.5, 0.999, 1e-8, step)
    torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-6, atol=1e-7)


def test_bad_arguments_are_rejected(cuda):
    from building_gan_amd import _lib
    lib = _lib.load()
    assert lib.bg_spmm_f32(None, 4, None, None, None, None, None, 0, None, 0, None, 4, 10, 4, None) == -1
    assert lib.bg_cross_logits_f32(1, 100, 1, 100, 1, 1, 1, 1, 10, 100, None) == -2   # H % 64 != 0
    with pytest.raises(_lib.BganError):
        from building_gan_amd import ops
        ops.gemm_raw([torch.zeros(4, 4)], torch.zeros(4, 4), False, True)            # CPU tensors: no fallback


def test_bad_arguments_of_the_generator_and_critic_nodes_are_rejected(cuda):
    """The entry points added for the one-node generator / critic paths refuse what they cannot run, before
    any launch: missing operands (BG_ERR_ARG = -1), unsupported widths (BG_ERR_SHAPE = -2); the Python
    wrappers raise BganError for plans / shapes that do not fit."""
    from building_gan_amd import _lib, ops
    import copy
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(64, 128, device=cuda)
    i32 = torch.zeros(65, dtype=torch.int32, device=cuda)
    # story backward: a missing output, a width that is not a multiple of 4
    assert lib.bg_story_bwd_f32(x.data_ptr(), 128, x.data_ptr(), 128, x.data_ptr(), 128, i32.data_ptr(), None,
                                x.data_ptr(), 0.01, None, 128, x.data_ptr(), 128, x.data_ptr(), 128, 1, 128, st) == -1
    assert lib.bg_story_bwd_f32(x.data_ptr(), 128, x.data_ptr(), 128, x.data_ptr(), 128, i32.data_ptr(), None,
                                x.data_ptr(), 0.01, x.data_ptr(), 128, x.data_ptr(), 128, x.data_ptr(), 128, 1, 126,
                                st) == -2
    # gathered addend without its row index; accumulator start with too small a pitch
    assert lib.bg_spmm_add_f32(x.data_ptr(), 128, i32.data_ptr(), None, None, 0, None, 0, None, 0, None, 0,
                               x.data_ptr(), 128, None, None, 0, x.data_ptr(), 128, 64, 128, st) == -1
    g_, p = _plan(copy.deepcopy(_batch("tiny", 12, False)), cuda)
    with pytest.raises(_lib.BganError):          # story segments that do not cover the operand's rows
        ops.story_bwd_raw(p.story_seg, x[:8], x[:8], x[:8], 0.01)
    with pytest.raises(_lib.BganError):          # accumulator start of the wrong shape
        ops.gemm_raw([x], torch.zeros(32, 128, device=cuda), False, True, acc_init=torch.zeros(64, 16, device=cuda))
    with pytest.raises(_lib.BganError):          # row-lookup scatter with a segment map of the wrong length
        ops.scatter_arg_raw(x[:4], torch.zeros(4, 128, dtype=torch.int32, device=cuda), 64,
                            seg_of=torch.zeros(3, dtype=torch.int32, device=cuda))
    # coefficient inputs: the static column block must hold features + levels + 4 position columns + the one
    assert lib.bg_d_coef_init_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                  4 * 128 + 4, x.data_ptr(), 128, 4, 6, 20, 28, 8, x.data_ptr(), x.data_ptr(), st) == -1


def test_mask_gumbel_two_way(cuda):
    """Fused mask kernel against the reference expression (models.py:109-111, util.py:6-10), with its gradient."""
    from building_gan_amd import ops
    from oracle.pointer import softmax_to_hard
    torch.manual_seed(3)
    n = 5000
    logits = (torch.randn(n, 2) * 3).requires_grad_(True)
    expn = torch.empty(n, 2).exponential_()
    soft = (logits - expn.log()).softmax(-1)
    hard = softmax_to_hard(soft, -1)
    lg = logits.detach().to(cuda).requires_grad_(True)
    s0, h0 = ops.MaskGumbel.apply(lg, expn.to(cuda))
    torch.testing.assert_close(s0.cpu(), soft[:, :1].detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(h0.cpu(), hard[:, :1].detach(), rtol=0, atol=2e-7)
    assert torch.equal(h0.cpu().round(), hard[:, :1].detach().round())           # same winners
    w = torch.randn(n, 1)
    (soft[:, :1] * w + 2 * hard[:, :1] * w).sum().backward()
    (s0 * w.to(cuda) + 2 * h0 * w.to(cuda)).sum().backward()
    torch.testing.assert_close(lg.grad.cpu(), logits.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n_seg,H", [(50, 128), (50, 256), (8, 256), (20, 320), (300, 128), (600, 128)])
def test_one_launch_segment_sum_and_max(cuda, n_seg, H):
    """chunk_ptr == NULL form (a workgroup per segment) against the oracle: many segments of moderate,
    ragged length, two of them empty, ties inside a segment.  Up to 512 segments a reduction runs 1024 threads per
    segment with a tree combine over the lane groups; above, 256 threads with the in-order combine."""
    from building_gan_amd import ops
    from building_gan_amd.plan import LinOp, SCALE_DIV, _Builder
    from oracle.scatter import scatter_max, scatter_mean
    torch.manual_seed(7)
    n = 4000 if n_seg >= 32 else 100 * n_seg              # (a few segments go one per workgroup only when short)
    seg = torch.randint(0, n_seg, (n,))
    seg[seg == 11] = 12
    seg[seg == 30] = 29                                   # segments 11 and 30 are empty
    seg[seg == 2] = 3                                     # and segment 2 (the 8-segment case has no 11 or 30)
    x = torch.rand(n, H) * 2 - 1
    first = int((seg == 5).nonzero()[0])
    later = int((seg == 5).nonzero()[-1])
    x[later] = x[first] = 3.0                             # a tie at the maximum: the first slot must win
    b = _Builder(cuda)
    c = b.csr("s", seg.to(cuda), torch.arange(n).to(cuda), n_seg, n, counts=True)
    b.maybe_chunks(c, force=True)
    c.max_len = int(torch.bincount(seg, minlength=n_seg).max())
    assert c.row_per_workgroup
    val, arg = ops.segmax_raw(c, x.to(cuda))
    rv, ra = scatter_max(x, seg, n_seg)
    assert torch.equal(val.cpu(), rv)
    assert torch.equal(arg.cpu().long(), torch.where(ra == n, torch.full_like(ra, -1), ra))
    mean = ops.spmm_raw(LinOp(c, c.count, SCALE_DIV), x.to(cuda))
    torch.testing.assert_close(mean.cpu(), scatter_mean(x, seg, n_seg), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,Ks,transB", [(200, 96, [64, 36], True),        # skinny kernel, 4 waves
                                            (500, 256, [256, 256], True),     # skinny kernel, whole-line loads (K_s % 32 == 0)
                                            (500, 256, [512], False),         # same, B stored [K, N]
                                            (301, 200, [32, 96, 160, 64], True),   # same, four segments, ragged M and N
                                            (301, 200, [32, 96, 160, 64], False),
                                            (70, 40, [1024, 512], True),      # same, more than two chunks per wave
                                            (200, 96, [260, 100], True),      # skinny kernel, 8 waves, fragment loads
                                            (200, 96, [260, 100], False),
                                            (11520, 128, [128, 128, 4], True),  # 64x64 register-staged kernel
                                            (11520, 128, [128, 128], True),     # row-panel kernel, 48-row panels (one per CU)
                                            (3001, 128, [64, 192], True),       # row-panel kernel, 32-row panels, ragged last panel
                                            (12577, 128, [128], True),          # row-panel kernel, 64-row panels
                                            (9000, 260, [128], False),        # same, B stored [K, N]
                                            (40000, 128, [128, 128, 4], True),  # LDS-DMA ring kernel (>= 512 tiles)
                                            (33000, 104, [256], False)])      # LDS-DMA ring kernel, B stored [K, N]
def test_gemm_epilogue_all_kernels(cuda, M, N, Ks, transB):
    """Every GEMM kernel variant with the full epilogue: bias, row mask, gathered pre-activation addend
    (pre_add_rows), LeakyReLU, residual -- against float64."""
    from building_gan_amd import ops
    torch.manual_seed(M + N)
    K = sum(Ks)
    xs = [torch.rand(M, k) * 2 - 1 for k in Ks]
    W = torch.rand(N, K) * 2 - 1 if transB else torch.rand(K, N) * 2 - 1
    b = torch.rand(N) * 2 - 1
    rs = (torch.rand(M) > 0.2).float()
    G = 37
    term = torch.rand(G, N) * 2 - 1
    rows = torch.randint(0, G, (M,), dtype=torch.int32)
    res = torch.rand(M, N) * 2 - 1
    Wd = W.double().t() if transB else W.double()
    ref = (torch.cat(xs, 1).double() @ Wd + b.double()) * rs.double().view(-1, 1) + term.double()[rows.long()]
    ref = torch.nn.functional.leaky_relu(ref, 0.01) + res.double()
    out = ops.gemm_raw([x.to(cuda) for x in xs], W.to(cuda), False, transB, bias=b.to(cuda), row_scale=rs.to(cuda),
                       act=ops.ACT_LEAKY, slope=0.01, residual=res.to(cuda), pre_add=term.to(cuda),
                       pre_rows=rows.to(cuda))
    scale = torch.cat(xs, 1).abs().double() @ Wd.abs()
    err = (out.cpu().double() - ref).abs()
    assert float((err / (scale + 1.0)).max()) < 2e-6


def test_splitk_small_output_reduce(cuda):
    """Weight gradient of a narrow layer: few outputs, hundreds of splits (k_splitk_reduce_small)."""
    from building_gan_amd import ops
    torch.manual_seed(9)
    R, M, N = 11520, 2, 64
    g, x = torch.rand(R, M) * 2 - 1, torch.rand(R, N) * 2 - 1
    out = ops.gemm_tn_raw(g.to(cuda), x.to(cuda))
    ref = g.double().t() @ x.double()
    assert float((out.cpu().double() - ref).abs().max()) < 1e-3


def _cheb_dense(A, L):
    """[T_0(A), .., T_L(A)] in fp64 (dense)."""
    T = [torch.eye(A.shape[0], dtype=torch.float64), A]
    for k in range(2, L + 1):
        T.append(2 * A @ T[-1] - T[-2])
    return T[:L + 1]


@pytest.mark.parametrize("basis", [0, 1])
@pytest.mark.parametrize("preset,ragged,W,L", [("small", True, 8, 5), ("dataset", False, 32, 12), ("dataset", False, 24, 12),
                                               ("tiny", False, 4, 1), ("tiny", False, 4, 2)])
def test_operator_chain_matches_separate_launches(cuda, preset, ragged, W, L, basis):
    """bg_spmm_chain_f32 (a workgroup per graph, rows resident in LDS between steps) against the same steps
    as separate bg_spmm_f32 / bg_spmm_add_f32 launches: same slot order and scale arithmetic, so the
    mean-aggregation powers are bit-exact; and against the dense fp64 operator.  basis 1: the three-term (Chebyshev /
    Clenshaw) forms of the same kernels against the same recurrences run launch by launch, and against T_k(A) in fp64."""
    from building_gan_amd import ops
    g, p = _plan(_batch(preset, 3, ragged), cuda)
    blocks = p.vox_blocks
    assert blocks is not None and blocks.n == p.B
    gen = torch.Generator().manual_seed(5)
    N = p.Nv
    x = (torch.rand(N, W, generator=gen) * 2 - 1).to(cuda)
    import dataclasses
    assert blocks.max_deg <= 6                                   # grid graphs: the register-index kernel, 6 slots
    wide = dataclasses.replace(blocks, max_deg=8)                # same plan through the 8-slot variant ..
    general = dataclasses.replace(blocks, max_deg=1 << 20)       # .. and through the LDS-index kernel
    for op in (p.vox_mean, p.vox_mean.T):
        for blk in (blocks, wide, general):
            # powers, written into a wider buffer (row pitch > (L+1)*W exercises the pitch arguments)
            Z = torch.zeros(N, (L + 1) * W + 4, device=cuda)
            Z[:, :W] = x
            Zref = Z.clone()
            ops.power_chain_raw(op, Z, L, W, blk, basis)
            ops.power_chain_raw(op, Zref, L, W, None, basis)       # L separate launches
            assert torch.equal(Z, Zref)
            # Horner / Clenshaw form with addends
            Y = (torch.rand(N, L + 1, W, generator=gen) * 2 - 1).to(cuda)
            acc = ops.horner_chain_raw(op, Y, L, W, blk, basis)
            ref = ops.horner_chain_raw(op, Y, L, W, None, basis)
            assert torch.equal(acc, ref)
    # dense fp64 statement of the mean-aggregation operator (oracle: scatter-mean over in-edges)
    ve = g.voxel_edge.cpu()
    A = torch.zeros(N, N, dtype=torch.float64)
    A.index_put_((ve[1], ve[0]), torch.ones(ve.shape[1], dtype=torch.float64), accumulate=True)
    A = A / A.sum(1, keepdim=True).clamp(min=1)
    Z = torch.zeros(N, L + 1, W, device=cuda)
    Z[:, 0] = x
    ops.power_chain_raw(p.vox_mean, Z, L, W, blocks, basis)
    vecs = [x.double().cpu()]                                     # p_k(A) x by the recurrence on vectors
    for k in range(1, L + 1):
        nxt = A @ vecs[-1]
        vecs.append(2 * nxt - vecs[-2] if (basis and k >= 2) else nxt)
        np.testing.assert_allclose(Z[:, k].cpu().numpy(), vecs[k].numpy(), rtol=RTOL, atol=ATOL)
    if N <= 4096:       # sum_k p_k(A^T) Y_k against the dense polynomials
        polys = _cheb_dense(A, L) if basis else [torch.linalg.matrix_power(A, k) for k in range(L + 1)]
        Y = (torch.rand(N, L + 1, W, generator=gen) * 2 - 1)
        acc = ops.horner_chain_raw(p.vox_mean.T, Y.to(cuda), L, W, blocks, basis)
        ref = sum(polys[k].t() @ Y[:, k].double() for k in range(L + 1))
        np.testing.assert_allclose(acc.cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL * (L + 1))


@pytest.mark.parametrize("basis", [0, 1])
def test_operator_chain_three_rows_per_thread(cuda, basis):
    """Graphs of 2,049..3,072 voxels (the jittered dataset stream has them: 14 x 14 x 11 = 2,156) stay in the register-index
    chain kernel with three rows per thread: bit-exact against the separate launches and against the LDS-index kernel
    (which 8-slot plans of that size take), powers / Horner / pair forms, both bases."""
    from building_gan_amd import ops
    import dataclasses
    g, p = _plan(_batch("dataset", 2, "jitter"), cuda)
    blocks = p.vox_blocks
    assert 2048 < blocks.max_rows <= 3072 and blocks.max_deg <= 6
    gen = torch.Generator().manual_seed(8)
    N, W, L = p.Nv, 8, 12
    x = (torch.rand(N, W, generator=gen) * 2 - 1).to(cuda)
    Y = (torch.rand(N, L + 1, W, generator=gen) * 2 - 1).to(cuda)
    general = dataclasses.replace(blocks, max_deg=8)                 # 8 slots x 3 rows would spill: the LDS-index kernel
    for op in (p.vox_mean, p.vox_mean.T):
        outs = []
        for blk in (blocks, general, None):
            Z = torch.zeros(N, (L + 1) * W, device=cuda)
            Z[:, :W] = x
            ops.power_chain_raw(op, Z, L, W, blk, basis)
            outs.append((Z, ops.horner_chain_raw(op, Y, L, W, blk, basis)))
        for Z, acc in outs[:2]:
            assert torch.equal(Z, outs[2][0]) and torch.equal(acc, outs[2][1])
    Z1 = torch.zeros(N, (L + 1) * W, device=cuda); Z1[:, :W] = x
    Z2 = torch.zeros(N, (L + 1) * 4, device=cuda); Z2[:, :4] = x[:, :4]
    R1, R2 = Z1.clone(), Z2.clone()
    ops.power_chain_pair_raw(p.vox_mean, Z1, W, Z2, 4, L, blocks, basis)
    ops.power_chain_raw(p.vox_mean, R1, L, W, None, basis)
    ops.power_chain_raw(p.vox_mean, R2, L, 4, None, basis)
    assert torch.equal(Z1, R1) and torch.equal(Z2, R2)


@pytest.mark.parametrize("basis", [0, 1])
def test_operator_chain_autograd_pair(cuda, basis):
    """PowerChain / HornerChain are each other's backward (in either basis): gradients against a dense fp64 statement."""
    from building_gan_amd import ops
    g, p = _plan(_batch("small", 4, True), cuda)
    N, W, L = p.Nv, 8, 4
    ve = g.voxel_edge.cpu()
    A = torch.zeros(N, N, dtype=torch.float64)
    A.index_put_((ve[1], ve[0]), torch.ones(ve.shape[1], dtype=torch.float64), accumulate=True)
    A = A / A.sum(1, keepdim=True).clamp(min=1)
    gen = torch.Generator().manual_seed(6)
    x = (torch.rand(N, W, generator=gen) * 2 - 1)
    Gz = (torch.rand(N, L + 1, W, generator=gen) * 2 - 1)
    xd = x.double().requires_grad_(True)
    polys = _cheb_dense(A, L) if basis else [torch.linalg.matrix_power(A, k) for k in range(L + 1)]
    Zd = torch.stack([polys[k] @ xd for k in range(L + 1)], dim=1)
    (Zd * Gz.double()).sum().backward()
    xg = x.to(cuda).requires_grad_(True)
    Z = ops.PowerChain.apply(xg, p.vox_mean, L, p.vox_blocks, basis)
    (Z * Gz.to(cuda)).sum().backward()
    np.testing.assert_allclose(Z.detach().cpu().numpy(), Zd.detach().numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xd.grad.numpy(), rtol=RTOL, atol=ATOL)
    # Horner over the transposed operator, gradient with respect to every addend
    Yd = Gz.double().requires_grad_(True)
    out_d = sum(polys[k].t() @ Yd[:, k] for k in range(L + 1))
    (out_d * x.double()).sum().backward()
    Yg = Gz.to(cuda).requires_grad_(True)
    out = ops.HornerChain.apply(Yg, p.vox_mean.T, p.vox_blocks, basis)
    (out * x.to(cuda)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_d.detach().numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(Yg.grad.cpu().numpy(), Yd.grad.numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("basis", [0, 1])
@pytest.mark.parametrize("preset,ragged", [("small", True), ("dataset", False)])
def test_operator_chain_pair_equals_two_chains(cuda, preset, ragged, basis):
    """bg_spmm_chain_pair_f32 (the powers of two matrices of different width and pitch in ONE launch) against
    two bg_spmm_chain_f32 launches: identical bits, for both register-index variants; a plan outside the
    short-row kernel falls back to the two launches."""
    from building_gan_amd import ops
    import dataclasses
    g, p = _plan(_batch(preset, 13, ragged), cuda)
    gen = torch.Generator().manual_seed(29)
    N, L, W1, W2 = p.Nv, 12, 32, 24
    x1 = (torch.rand(N, W1, generator=gen) * 2 - 1).to(cuda)
    x2 = (torch.rand(N, W2, generator=gen) * 2 - 1).to(cuda)
    for blk in (p.vox_blocks, dataclasses.replace(p.vox_blocks, max_deg=8),
                dataclasses.replace(p.vox_blocks, max_deg=1 << 20)):
        Z1 = torch.zeros(N, (L + 1) * W1, device=cuda); Z1[:, :W1] = x1
        Z2 = torch.zeros(N, (L + 1) * W2 + 8, device=cuda); Z2[:, :W2] = x2
        R1, R2 = Z1.clone(), Z2.clone()
        ops.power_chain_pair_raw(p.vox_mean, Z1, W1, Z2, W2, L, blk, basis)
        ops.power_chain_raw(p.vox_mean, R1, L, W1, blk, basis)
        ops.power_chain_raw(p.vox_mean, R2, L, W2, blk, basis)
        assert torch.equal(Z1, R1) and torch.equal(Z2, R2)
        assert float(Z1[:, -W1:].abs().max()) > 0 and float(Z2[:, L * W2:(L + 1) * W2].abs().max()) > 0


@pytest.mark.parametrize("basis", [0, 1])
@pytest.mark.parametrize("preset,ragged", [("small", True), ("dataset", False), ("dataset", "jitter")])
def test_operator_chain_on_column_groups_equals_regrouped_operands(cuda, preset, ragged, basis):
    """bg_spmm_chain_grouped_f32: a chain whose W = P w columns are P groups living in P separate [N, K w] matrices (the
    layout a product over P N stacked rows leaves) against bg_regroup_f32 + the plain chain (+ bg_regroup_f32 back): bit
    for bit, Horner / Clenshaw form reading groups and the powers form writing them, all three kernel variants."""
    from building_gan_amd import _lib, ops
    import dataclasses
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g, p = _plan(_batch(preset, 17, ragged), cuda)
    gen = torch.Generator().manual_seed(31)
    N, L, P, w = p.Nv, 12, 2, 8
    K1, W = L + 1, P * w
    X = (torch.rand(P * N, K1 * w, generator=gen) * 2 - 1).to(cuda)          # what the product leaves: [P][N, K1 w]
    d0 = (torch.rand(N, W, generator=gen) * 2 - 1).to(cuda)

    def chain(op, blk, x0, ldx, add, ld_add, add_step, out, ldo, out_step, bas, grp):
        c = op.csr
        fn = lib.bg_spmm_chain_grouped_f32
        rc = fn(x0, ldx, c.rowptr.data_ptr(), c.col.data_ptr(), op.row_scale.data_ptr() if op.row_scale is not None else None,
                op.row_mode, op.col_scale.data_ptr() if op.col_scale is not None else None, op.col_mode, blk.ptr.data_ptr(),
                blk.n, blk.max_rows, blk.max_nnz, blk.max_deg, add, ld_add, add_step, out, ldo, out_step, L, W, bas, *grp, st)
        assert rc == 0, rc
    f = 4
    half = N * K1 * w
    for blk in (p.vox_blocks, dataclasses.replace(p.vox_blocks, max_deg=8), dataclasses.replace(p.vox_blocks, max_deg=1 << 20)):
        if blk.max_deg == 8 and blk.max_rows > 2048:
            continue                                                    # (8 slots x 3 rows: that plan takes the LDS-index kernel anyway)
        # ---- Horner / Clenshaw over the transposed operator, operands read in groups
        Ys = torch.empty(N, K1, P, w, device=cuda)
        assert lib.bg_regroup_f32(X.data_ptr(), Ys.data_ptr(), N, K1, P, w, 0, st) == 0
        assert torch.equal(Ys, X.view(P, N, K1, w).permute(1, 2, 0, 3))
        ref = ops.horner_chain_raw(p.vox_mean.T, Ys.view(N, K1, W), L, W, blk, basis)
        acc = torch.full((N, W), float("nan"), device=cuda)
        chain(p.vox_mean.T, blk, X.data_ptr() + f * L * w, K1 * w, X.data_ptr() + f * (L - 1) * w, K1 * w, -w,
              acc.data_ptr(), W, 0, 2 if basis else 0, (w, half, half, 0))
        assert torch.equal(acc, ref)
        # ---- powers, results written in groups (block 0 of the grouped matrix is not the chain's to write)
        Z = torch.zeros(N, K1 * W, device=cuda)
        Z[:, :W] = d0
        ops.power_chain_raw(p.vox_mean, Z, L, W, blk, basis)
        want = torch.empty(P * N, K1 * w, device=cuda)
        assert lib.bg_regroup_f32(Z.data_ptr(), want.data_ptr(), N, K1, P, w, 1, st) == 0
        got = torch.full((P * N, K1 * w), float("nan"), device=cuda)
        chain(p.vox_mean, blk, d0.data_ptr(), W, None, 0, 0, got.data_ptr() + f * w, K1 * w, w, 1 if basis else 0, (w, 0, 0, half))
        assert torch.equal(got[:, w:], want[:, w:]) and bool(torch.isnan(got[:, :w]).all())
    # argument checks: group width must divide W and be a multiple of 4; steps without a width are refused
    blk = p.vox_blocks
    c = p.vox_mean.csr
    bad = lambda *grp: lib.bg_spmm_chain_grouped_f32(
        d0.data_ptr(), W, c.rowptr.data_ptr(), c.col.data_ptr(), p.vox_mean.row_scale.data_ptr(), p.vox_mean.row_mode, None, 0,
        blk.ptr.data_ptr(), blk.n, blk.max_rows, blk.max_nnz, blk.max_deg, None, 0, 0, got.data_ptr(), K1 * w, w, L, W, 0, *grp, st)
    assert bad(6, 0, 0, half) == -2 and bad(12, 0, 0, half) == -2 and bad(0, 0, 0, half) == -1 and bad(w, 0, 0, half + 2) == -3


def test_operator_chain_rejects_oversized_blocks(cuda):
    """A diagonal block that cannot live in LDS is refused (BG_ERR_SHAPE); the host then runs the steps one by one."""
    from building_gan_amd import _lib, ops
    from building_gan_amd.plan import VoxelBlocks
    g, p = _plan(_batch("small", 4, True), cuda)
    big = VoxelBlocks(p.vox_blocks.ptr, p.vox_blocks.n, 1 << 16, 1 << 18)
    assert not ops._chain_fits(p.vox_mean, big, 4, torch.zeros(p.Nv, 8, device=cuda))
    lib = _lib.load()
    x = torch.zeros(p.Nv, 8, device=cuda)
    c = p.vox_mean.csr
    rc = lib.bg_spmm_chain_f32(x.data_ptr(), 8, c.rowptr.data_ptr(), c.col.data_ptr(), None, 0, None, 0,
                               big.ptr.data_ptr(), big.n, big.max_rows, big.max_nnz, 6, None, 0, 0,
                               x.data_ptr() + 16, 8, 0, 1, 4, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == -2
    ok = p.vox_blocks
    rc = lib.bg_spmm_chain_f32(x.data_ptr(), 8, c.rowptr.data_ptr(), c.col.data_ptr(), None, 0, None, 0,
                               ok.ptr.data_ptr(), ok.n, ok.max_rows, ok.max_nnz, 6, None, 0, 0,
                               x.data_ptr() + 16, 8, 0, 1, 4, 3, torch.cuda.current_stream().cuda_stream)
    assert rc == -1                                                   # an unknown basis is an argument error


def test_chebyshev_shift_and_its_adjoint(cuda):
    """bg_cheb_shift_f32: multiplication by x on a table of Chebyshev coefficients against the polynomial identity
    (numpy.polynomial.chebyshev.chebmulx) and <shift(X), G> == <X, shift^T(G)> for the transposed form."""
    from building_gan_amd import ops
    gen = torch.Generator().manual_seed(41)
    K, blk = 7, 24
    X = torch.randn(K, blk, generator=gen)
    X[-1] = 0                                                         # a degree K-2 polynomial: x * p fits K blocks
    G = torch.randn(K, blk, generator=gen)
    out = torch.full((K, blk), float("nan"), device=cuda)
    ops.cheb_shift_raw(X.to(cuda), out, K, K, blk, False)
    ref = np.stack([np.polynomial.chebyshev.chebmulx(X[:-1, j].double().numpy()) for j in range(blk)], axis=1)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    part = torch.full((K, blk), 7.0, device=cuda)                     # K_out < K leaves the other blocks alone
    ops.cheb_shift_raw(X.to(cuda), part, K, 3, blk, False)
    assert torch.equal(part[:3], out[:3]) and bool((part[3:] == 7.0).all())
    Xf = torch.randn(K, blk, generator=gen)                           # the adjoint pairing holds for any table
    sh = torch.empty(K, blk, device=cuda); sht = torch.empty(K, blk, device=cuda)
    ops.cheb_shift_raw(Xf.to(cuda), sh, K, K, blk, False)
    ops.cheb_shift_raw(G.to(cuda), sht, K, K, blk, True)
    lhs = float((sh.double().cpu() * G.double()).sum()); rhs = float((Xf.double() * sht.double().cpu()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_gp_sample_matches_reference_formula(cuda):
    """bg_gp_sample_f32 against the reference's elementwise chain (losses.py:76-79) on the CPU: bit-exact,
    including void voxels (all-zero rows -> sample 0) and exact ties (first index wins)."""
    from building_gan_amd import ops
    from oracle.pointer import softmax_to_hard
    gen = torch.Generator().manual_seed(9)
    n = 5000
    label = torch.zeros(n, 6)
    cls = torch.randint(0, 7, (n,), generator=gen)
    label[torch.arange(n)[cls < 6], cls[cls < 6]] = 1.0          # class 6 = void: all-zero row
    out = torch.rand(n, 6, generator=gen) * (torch.rand(n, 1, generator=gen) > 0.3)   # masked rows are zero
    out[:50] = 0.25                                               # ties
    label[:50] = 0.0
    u = torch.rand(n, 1, generator=gen)
    mixed = label * u + out * (1 - u)
    nonzero = (mixed.max(dim=-1)[0] != 0).type(torch.float32).view(-1, 1)
    ref = softmax_to_hard(mixed, -1) * nonzero
    got = ops.gp_sample_raw(label.to(cuda), out.to(cuda), u.to(cuda))
    assert torch.equal(got.cpu(), ref)


def test_pack_columns(cuda):
    from building_gan_amd import ops
    gen = torch.Generator().manual_seed(10)
    srcs = [torch.rand(777, 6, generator=gen) for _ in range(3)]
    dst = torch.full((777, 13 * 24), -1.0, device=cuda)
    ops.pack_columns_raw([s.to(cuda) for s in srcs], 8, dst)
    ref = torch.cat([torch.nn.functional.pad(s, (0, 2)) for s in srcs], dim=1)
    assert torch.equal(dst[:, :24].cpu(), ref)
    assert bool((dst[:, 24:] == -1.0).all())


def test_construct_output_fused_matches_oracle(cuda):
    """bg_construct_output_f32 (no autograd graph) against the oracle's construct_output: labels within fp32
    rounding of the scatter sums, arg-max program index bit-exact (ties and masked voxels: first edge)."""
    from building_gan_amd import ops
    from oracle.networks import Attention as OA
    cpu = _batch("small", 11, True)
    import copy
    g, p = _plan(copy.deepcopy(cpu), cuda)
    gen = torch.Generator().manual_seed(11)
    Nv, Ec = p.Nv, p.Ec
    mask_hard = (torch.rand(Nv, 1, generator=gen) > 0.4).float()
    att_soft = torch.rand(Ec, 1, generator=gen)
    cev = cpu.cross_edge_voxel_index_select
    win = torch.zeros(Ec, 1)
    first = torch.zeros(Nv, dtype=torch.long).scatter_reduce_(0, cev, torch.arange(Ec), "amin", include_self=False)
    last = torch.zeros(Nv, dtype=torch.long).scatter_reduce_(0, cev, torch.arange(Ec), "amax", include_self=False)
    pick = first + (torch.rand(Nv, generator=gen) * (last - first + 1).float()).long().clamp(max=10 ** 9)
    pick = torch.minimum(pick, last)
    win[pick] = 1.0
    att_hard = (win - att_soft) + att_soft                      # straight-through values, not exactly 1.0
    att = {"hard": att_hard, "soft": att_soft}
    mask = {"hard": mask_hard, "soft": mask_hard}
    ro = OA.construct_output(cpu.program_class_feature, Nv, att, mask, cpu.cross_edge_program_index_select, cev)
    out, soft, idx = ops.construct_output_raw(mask_hard.to(cuda), att_hard.to(cuda), att_soft.to(cuda), p,
                                              g.program_class_feature)
    np.testing.assert_allclose(out.cpu().numpy(), ro[0].numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(soft.cpu().numpy(), ro[1].numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(idx.cpu(), ro[2])


def test_critic_loss_node(cuda):
    """ops.CriticLoss against the stock formula: value, penalty and all three gradients."""
    from building_gan_amd import ops
    gen = torch.Generator().manual_seed(12)
    B, S, N = 8, 80, 3000
    vb = torch.randn(2 * B, 1, generator=gen, dtype=torch.float64)
    vs = torch.randn(2 * S, 1, generator=gen, dtype=torch.float64)
    gr = torch.randn(N, 2, 8, generator=gen, dtype=torch.float64)
    gr[:, :, 6:] = 0
    gr[5] = 0                                                     # a zero-norm row: gradient 0, value (0-1)^2
    wb = torch.cat([torch.full((B,), -1.0 / B), torch.full((B,), 1.0 / B)])
    ws = torch.cat([torch.full((S,), -1.0 / S), torch.full((S,), 1.0 / S)])
    coef = 10.0 / N
    for t in (vb, vs, gr):
        t.requires_grad_(True)
    pen_ref = ((gr.norm(2, 2) - 1) ** 2).sum()
    ref = torch.dot(vb.view(-1), wb.double()) + torch.dot(vs.view(-1), ws.double()) + coef * pen_ref
    (3.0 * ref).backward()
    hb, hs, hg = (t.detach().float().to(cuda).requires_grad_(True) for t in (vb, vs, gr))
    loss, pen = ops.CriticLoss.apply(hb, hs, hg, wb.to(cuda), ws.to(cuda), coef)
    (3.0 * loss).backward()
    np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=1e-5)
    np.testing.assert_allclose(float(pen), float(pen_ref.detach()), rtol=1e-5)
    np.testing.assert_allclose(hb.grad.cpu().numpy(), vb.grad.numpy(), rtol=1e-6)
    np.testing.assert_allclose(hs.grad.cpu().numpy(), vs.grad.numpy(), rtol=1e-6)
    np.testing.assert_allclose(hg.grad.cpu().numpy(), gr.grad.numpy(), rtol=1e-4, atol=1e-8)
    # without a penalty
    loss2, _ = ops.CriticLoss.apply(hb.detach(), hs.detach(), None, wb.to(cuda), ws.to(cuda), 0.0)
    np.testing.assert_allclose(float(loss2), float(torch.dot(vb.view(-1), wb.double()) + torch.dot(vs.view(-1), ws.double())),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("preset,ragged,H", [("small", True, 128), ("dataset", False, 128), ("tiny", False, 64)])
def test_gather_story_one_launch_equals_two(cuda, preset, ragged, H):
    """bg_gather_story_f32 against bg_spmm_f32 + bg_story_project_f32 on the same operand: identical bits."""
    from building_gan_amd import ops
    g, p = _plan(_batch(preset, 7, ragged), cuda)
    gen = torch.Generator().manual_seed(13)
    x = (torch.rand(p.Nv, H, generator=gen) * 2 - 1).to(cuda)
    W = (torch.rand(H, 2 * H, generator=gen) * 2 - 1).to(cuda)
    gm, pq = ops.gather_story_raw(p.vox_mean, p.story_seg, x, W)
    assert torch.equal(gm, ops.spmm_raw(p.vox_mean, x))
    assert torch.equal(pq, ops.story_project_raw(p.story_seg, x, W))


@pytest.mark.parametrize("preset,ragged,H", [("small", True, 128), ("dataset", False, 128), ("tiny", False, 64),
                                             ("small", True, 256)])
def test_story_backward_one_launch(cuda, preset, ragged, H):
    """bg_story_bwd_f32 against the three steps it replaces: the leaky ReLU's backward (bit-exact: one multiply),
    the per-story sums of it and the per-story means of v (fp64 statement; the summation order differs)."""
    from building_gan_amd import ops
    import copy
    cpu = _batch(preset, 9, ragged)
    g_, p = _plan(copy.deepcopy(cpu), cuda)
    gen = torch.Generator().manual_seed(17)
    g = (torch.rand(p.Nv, H, generator=gen) * 2 - 1)
    y = (torch.rand(p.Nv, H, generator=gen) * 2 - 1)
    y[::7] = 0.0                                                   # y == 0 takes the slope branch, as aten's does
    v = (torch.rand(p.Nv, H, generator=gen) * 2 - 1)
    gz, gterm, story = ops.story_bwd_raw(p.story_seg, g.to(cuda), y.to(cuda), v.to(cuda), 0.01)
    ref = torch.ops.aten.leaky_relu_backward(g, y, 0.01, True)
    assert torch.equal(gz.cpu(), ref)
    vfc = cpu.voxel_floor_cluster.long()
    S = p.S
    sums = torch.zeros(S, H, dtype=torch.float64).index_add_(0, vfc, ref.double())
    cnt = torch.zeros(S, dtype=torch.float64).index_add_(0, vfc, torch.ones(p.Nv, dtype=torch.float64)).clamp(min=1)
    means = torch.zeros(S, H, dtype=torch.float64).index_add_(0, vfc, v.double()) / cnt.view(-1, 1)
    assert float((gterm.cpu().double() - sums).abs().max()) <= 1e-5 * float(sums.abs().max())
    assert float((story.cpu().double() - means).abs().max()) <= 1e-6
    # and the separate launches agree with it to rounding
    assert float((gterm - ops.spmm_raw(p.story_gather.T, gz)).abs().max()) <= 1e-5 * float(sums.abs().max())
    assert float((story - ops.spmm_raw(p.story_mean, v.to(cuda))).abs().max()) <= 1e-6


@pytest.mark.parametrize("H", [128, 6])
def test_spmm_add_with_gathered_addend(cuda, H):
    """bg_spmm_add_f32's gathered addend: op(x) + add1 + add2 + gadd[rows] / scale[rows] in one launch equals the
    plain launch followed by the gather, the division and the addition in the same order (bit-exact), on the
    float4 path (H = 128) and on the scalar path (H = 6)."""
    from building_gan_amd import ops
    from building_gan_amd.plan import SCALE_DIV
    import copy
    g_, p = _plan(copy.deepcopy(_batch("small", 10, True)), cuda)
    gen = torch.Generator().manual_seed(19)
    x, a1, a2 = [(torch.rand(p.Nv, H, generator=gen) * 2 - 1).to(cuda) for _ in range(3)]
    gs = (torch.rand(p.S, H, generator=gen) * 2 - 1).to(cuda)
    rows, cnt = p.story_gather.csr.col, p.story_seg.count
    base = ops.spmm_add_raw(p.vox_mean.T, x, a1, a2)
    got = ops.spmm_add_raw(p.vox_mean.T, x, a1, a2, gather=(gs, rows, cnt, SCALE_DIV))
    assert torch.equal(got, base + gs[rows.long()] / cnt[rows.long()].view(-1, 1))
    got = ops.spmm_add_raw(p.vox_mean.T, x, a1, None, gather=(gs, rows, None, 0))
    assert torch.equal(got, ops.spmm_add_raw(p.vox_mean.T, x, a1) + gs[rows.long()])


def test_scatter_arg_row_lookup_equals_fill_and_scatter(cuda):
    """bg_scatter_arg_f32 with the rows' segment ids (one launch, every row looks its value up) against the zero
    fill + scatter form, for the voxel -> story, voxel -> graph and story -> graph pooling plans: identical."""
    from building_gan_amd import ops
    import copy
    g_, p = _plan(copy.deepcopy(_batch("small", 11, True)), cuda)
    gen = torch.Generator().manual_seed(23)
    H = 64
    for csr, n in ((p.story_seg, p.Nv), (p.graph_seg, p.Nv), (p.story_graph_seg, p.S)):
        assert csr.seg_of is not None and csr.seg_of.numel() == n
        x = (torch.rand(n, H, generator=gen) * 2 - 1).to(cuda)
        _, arg = ops.segmax_raw(csr, x)
        g = (torch.rand(csr.n_rows, H, generator=gen) * 2 - 1).to(cuda)
        want = ops.scatter_arg_raw(g, arg, n)
        assert torch.equal(ops.scatter_arg_raw(g, arg, n, seg_of=csr.seg_of), want)
        assert float(want.abs().sum()) > 0


def test_segment_sum_added_to_an_accumulator_in_one_launch(cuda):
    """bg_segsum_acc_f32 (the program features' gradient accumulated over the Attention calls of the generator backward):
    the sums themselves against a float64 index_add (weighted by edge id and plain), and accumulator + sums against the same
    launch into zeros followed by an elementwise add -- identical bits.  Cross edges grouped by program node (long ragged
    rows, more than 256 segments' worth of work per launch) and the story plan."""
    from building_gan_amd import ops
    import copy
    g_, p = _plan(copy.deepcopy(_batch("small", 13, True)), cuda)
    gen = torch.Generator().manual_seed(31)
    H = 128
    for csr, weighted in ((p.cross_p, True), (p.cross_p, False), (p.story_seg, False)):
        x = (torch.rand(csr.n_cols, H, generator=gen) * 2 - 1).to(cuda)
        w = torch.rand(csr.nnz, generator=gen).to(cuda) if weighted else None
        sums = ops.segsum_acc_raw(csr, x, torch.zeros(csr.n_rows, H, device=cuda), w)
        rp, col = csr.rowptr.cpu().long(), csr.col.cpu().long()
        seg = torch.repeat_interleave(torch.arange(csr.n_rows), rp[1:] - rp[:-1])
        rows = x.cpu().double()[col]
        if weighted:
            rows = rows * w.cpu().double()[csr.perm.cpu().long()].view(-1, 1)
        want = torch.zeros(csr.n_rows, H, dtype=torch.float64).index_add_(0, seg, rows)
        torch.testing.assert_close(sums.cpu().double(), want, rtol=1e-5, atol=1e-5)
        acc0 = (torch.rand(csr.n_rows, H, generator=gen) * 2 - 1).to(cuda)
        got = ops.segsum_acc_raw(csr, x, acc0.clone(), w)
        assert torch.equal(got, acc0 + sums)


def test_pooling_of_several_label_sets_in_one_launch(cuda):
    """bg_segmax_batch_f32 / bg_scatter_arg_batch_f32 (the critic's real, fake and sample rows pooled by one launch per
    level, their backward scattered by one) against one call per set: values, arg-max ids and scattered rows identical,
    for the voxel -> story and the story -> graph plans, three sets of H = 256 and two of H = 128, a tie included."""
    from building_gan_amd import ops
    import copy
    g_, p = _plan(copy.deepcopy(_batch("small", 12, True)), cuda)
    gen = torch.Generator().manual_seed(29)
    for csr, n in ((p.story_seg, p.Nv), (p.story_graph_seg, p.S)):
        for H, nb in ((256, 3), (128, 2)):
            x = (torch.rand(nb, n, H, generator=gen) * 2 - 1).to(cuda)
            x[0, 1] = x[0, 0]                                         # a tie inside the first segment: first index wins
            val, arg = ops.segmax_batch_raw(csr, x)
            for b in range(nb):
                v1, a1 = ops.segmax_raw(csr, x[b])
                assert torch.equal(val[b], v1) and torch.equal(arg[b], a1)
            g = (torch.rand(nb, csr.n_rows, H, generator=gen) * 2 - 1).to(cuda)
            got = ops.scatter_arg_batch_raw(g, arg, n, csr.seg_of)
            for b in range(nb):
                assert torch.equal(got[b], ops.scatter_arg_raw(g[b], arg[b], n, seg_of=csr.seg_of))
            assert float(got.abs().sum()) > 0
            total = torch.full((n, H), float("nan"), device=cuda)          # the sets' sum by the same launch, in set order
            again = ops.scatter_arg_batch_raw(g, arg, n, csr.seg_of, sum_out=total)
            want = got[0] + got[1] if nb == 2 else (got[0] + got[1]) + got[2]
            assert torch.equal(again, got) and torch.equal(total, want)
    lib_err = ops._lib.BganError
    with pytest.raises(lib_err):
        ops.segmax_batch_raw(p.story_seg, torch.zeros(2, p.Nv + 1, 8, device=cuda))


def test_pos_terms_and_static_columns(cuda):
    """bg_pos_terms_f32 against the scatter-mean statement (has * pos - mean_j pos_j: bit-exact, including an
    isolated voxel) and bg_d_static_cols_f32 against the concatenation it replaces."""
    from building_gan_amd import ops
    from oracle.scatter import scatter_mean
    cpu = _batch("small", 8, True)
    cpu.voxel_edge = cpu.voxel_edge[:, cpu.voxel_edge[1] != 3]          # voxel 3 loses its in-edges
    import copy
    g, p = _plan(copy.deepcopy(cpu), cuda)
    pos = cpu.voxel_feature[:, 3:6]
    ve = cpu.voxel_edge
    has = torch.zeros(p.Nv, 1).index_fill_(0, ve[1], 1.0)
    ref = has * pos - scatter_mean(pos[ve[0]], ve[1], p.Nv)
    got = ops.pos_terms_raw(p.vox_mean, g.voxel_feature[:, 3:6])
    assert torch.equal(got[:, :3].cpu(), ref) and bool((got[:, 3] == 0).all())
    assert float(got[3].abs().max()) == 0.0
    one = ops.pos_terms_raw(p.vox_mean, g.voxel_feature[:, 3:6], 1.0)       # pad column of ones (bias carrier)
    assert torch.equal(one[:, :3], got[:, :3]) and bool((one[:, 3] == 1).all())
    feat = torch.cat([cpu.voxel_feature[:, 0:3], cpu.voxel_feature[:, 6:]], dim=1)
    ML, ds = 20, 32
    Z = torch.full((p.Nv, 3 * ds), -7.0, device=cuda)
    ops.d_static_cols_raw(feat.to(cuda), p.level, ML, got, Z, ds)
    onehot = torch.nn.functional.one_hot(p.level.cpu().long(), ML).float()
    expect = torch.cat([feat, onehot, got.cpu(), torch.ones(p.Nv, 1), torch.zeros(p.Nv, ds - 4 - ML - 5)], dim=1)
    assert torch.equal(Z[:, :ds].cpu(), expect) and bool((Z[:, ds:] == -7.0).all())


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("K1,R,K,N,M", [(13, 40, 256, 256, 520), (13, 40, 256, 256, 80), (5, 12, 64, 96, 50), (2, 8, 32, 32, 16)])
def test_shift_on_load_product_equals_shift_then_product(cuda, K1, R, K, N, M, transpose):
    """bg_gemm_cheb_f32 (second operand = the Chebyshev shift of the first's table, formed while the product loads it)
    against bg_cheb_shift_f32 followed by bg_gemm_f32: bit for bit, for both maps, both layouts of B, partial row counts
    (the recurrence's early layers), with and without addends."""
    from building_gan_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator().manual_seed(47)
    X = (torch.rand(K1 * R, K, generator=gen) * 2 - 1).to(cuda)
    sh = torch.empty_like(X)
    ops.cheb_shift_raw(X, sh, K1, K1, R * K, transpose)
    for transB in (True, False):
        B = (torch.rand((N, 2 * K) if transB else (2 * K, N), generator=gen) * 2 - 1).to(cuda)
        for addends in (True, False):
            res = X[:M, :N].contiguous() if addends and N <= K else None
            pre = (torch.rand(M, N, generator=gen) * 2 - 1).to(cuda) if addends else None
            ref = ops.gemm_raw([X[:M], sh[:M]], B, False, transB, residual=res, pre_add=pre)
            out = torch.full((M, N), float("nan"), device=cuda)
            rc = lib.bg_gemm_cheb_f32(1 if transB else 0, M, N, X.data_ptr(), K, K, R, K1, 1 if transpose else 0, B.data_ptr(),
                                      B.stride(0), out.data_ptr(), N, res.data_ptr() if res is not None else None, N,
                                      pre.data_ptr() if pre is not None else None, N, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            assert torch.equal(out, ref), (transB, addends)
    # shapes outside the skinny kernel and aliasing are refused
    st = torch.cuda.current_stream().cuda_stream
    assert lib.bg_gemm_cheb_f32(1, M, N, X.data_ptr(), K, K, R, K1, 0, X.data_ptr(), 2 * K, X.data_ptr(), N, None, 0, None, 0, st) == -1
    big = torch.zeros(1, device=cuda)
    assert lib.bg_gemm_cheb_f32(1, 1 << 14, 1 << 10, X.data_ptr(), K, K, 1 << 14, 1, 0, X.data_ptr(), 2 * K, big.data_ptr(), 1 << 10,
                                None, 0, None, 0, st) == -2


def test_shift_of_stacked_tables_by_one_launch(cuda):
    """bg_cheb_shift_slabs_f32 (the shifts of all layers' states of the coefficient recurrence at once: slabs of one padding
    block + K coefficient blocks) against bg_cheb_shift_f32 per slab; padding blocks come out as zeros."""
    from building_gan_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator().manual_seed(48)
    L, du, H2 = 5, 12, 32
    K, P, blk = L + 1, L + 2, du * H2
    X = torch.randn(L + 1, P, du, H2, generator=gen).to(cuda)
    X[:, 0] = 0                                                      # the padding block of every slab
    out = torch.full((L * P + 1, du, H2), float("nan"), device=cuda)
    assert lib.bg_cheb_shift_slabs_f32(X.data_ptr(), out.data_ptr(), L * P + 1, P, 1, K, blk, 0,
                                       torch.cuda.current_stream().cuda_stream) == 0
    for m in range(L):
        ref = torch.empty(K, du, H2, device=cuda)
        ops.cheb_shift_raw(X[m, 1:].contiguous(), ref, K, K, blk, False)
        assert torch.equal(out[m * P + 1:(m + 1) * P], ref), m
        assert bool((out[m * P] == 0).all())
    assert bool((out[L * P] == 0).all())


@pytest.mark.parametrize("M", [3001, 11520, 12281, 12577])   # 32- and 48-row panels (ragged last one); 64-row panels: declined
@pytest.mark.parametrize("K", [[128, 128], [64]])             # four K steps / one (the index chain then ends after the loop)
def test_gemm_with_addend_formed_from_neighbour_rows(cuda, M, K):
    """bg_gemm_nbr_f32 (the gathered pre-activation addend formed on load: neighbour rows summed in slot order, IEEE
    division by the count, plus the row's own part) against bg_spmm_add_f32 followed by bg_gemm_f32: bit for bit, with
    rows of 0..5 neighbours (more than the two that travel through registers), accumulator start, LeakyReLU, residual and
    the pre-residual copy; shapes the row-panel kernel does not take are declined before anything is launched."""
    import ctypes as C
    from building_gan_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator().manual_seed(M + len(K))
    G, N = 83, 128
    R = lambda *s: (torch.rand(*s, generator=gen) * 2 - 1)
    deg = torch.randint(0, 6, (G,), generator=gen)
    deg[0], deg[1] = 0, 5
    rowptr = torch.zeros(G + 1, dtype=torch.int32); rowptr[1:] = deg.cumsum(0).to(torch.int32)
    col = torch.randint(0, G, (int(rowptr[-1]),), generator=gen).to(torch.int32)
    count = deg.clamp(min=1).float()
    pq = R(G, 2 * N).to(cuda)                                   # [own part | neighbour part], as the generator block keeps it
    rows = torch.sort(torch.randint(0, G, (M,), generator=gen))[0].to(torch.int32).to(cuda)
    xs = [R(M, k).to(cuda) for k in K]
    W, acc0 = R(N, sum(K)).to(cuda), R(M, N).to(cuda)
    rowptr, col, count = rowptr.to(cuda), col.to(cuda), count.to(cuda)
    st = torch.cuda.current_stream().cuda_stream
    term = torch.empty(G, N, device=cuda)
    assert lib.bg_spmm_add_f32(pq.data_ptr() + 4 * N, 2 * N, rowptr.data_ptr(), col.data_ptr(), count.data_ptr(), 2, None, 0,
                               pq.data_ptr(), 2 * N, None, 0, None, 0, None, None, 0, term.data_ptr(), N, G, N, st) == 0
    pro_ref = torch.empty(M, N, device=cuda)
    ref = ops.gemm_raw(xs, W, False, True, act=ops.ACT_LEAKY, slope=0.01, residual=xs[0][:, :N] if K[0] >= N else None,
                       pre_add=term, pre_rows=rows, acc_init=acc0, pre_residual_out=pro_ref)
    res = xs[0] if K[0] >= N else None
    pa, la, ka = (C.c_void_p * 4)(), (C.c_int64 * 4)(), (C.c_int64 * 4)()
    for i, x in enumerate(xs):
        pa[i], la[i], ka[i] = x.data_ptr(), x.stride(0), x.shape[1]
    out = torch.full((M, N), float("nan"), device=cuda)
    pro = torch.full((M, N), float("nan"), device=cuda)

    def call(m):
        return lib.bg_gemm_nbr_f32(m, N, len(xs), pa, la, ka, W.data_ptr(), W.stride(0), out.data_ptr(), N, None, ops.ACT_LEAKY, 0.01,
                                   res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0,
                                   pq.data_ptr(), 2 * N, pq.data_ptr() + 4 * N, 2 * N, rowptr.data_ptr(), col.data_ptr(),
                                   count.data_ptr(), rows.data_ptr(), acc0.data_ptr(), N, pro.data_ptr(), N, st)
    if M > 12288:                 # 64-row panels: the neighbour registers would cost the second workgroup per CU
        assert call(M) == -2 and bool(torch.isnan(out).all())
        return
    assert call(M) == 0
    assert torch.equal(out, ref) and torch.equal(pro, pro_ref)
    assert call(1000) == -2                                    # below the row-panel kernel's range: nothing launched



def test_fills_and_copies_as_one_launch(cuda):
    """bg_fill_copy_batch: zero fills and strided copies of different sizes and alignments in one launch, against torch;
    bytes outside every job keep their contents; bad arguments are refused."""
    import ctypes as C
    from building_gan_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(3)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.rand(1000, 260, generator=gen).to(cuda)           # destination of a fill: whole buffer, 16-byte aligned
    b = torch.rand(37, 50, generator=gen).to(cuda)              # fill of rows 3..30, columns 1..41 (4-byte aligned only)
    c_src = torch.rand(256, 1032, generator=gen).to(cuda)       # strided copy: a [256, 256] block of a wider matrix
    c_dst = torch.full((256, 256), -1.0, device=cuda)
    d_src = torch.rand(5, 7, generator=gen).to(cuda)            # small unaligned copy
    d_dst = torch.full((5, 9), -1.0, device=cuda)
    b_ref = b.clone(); b_ref[3:31, 1:42] = 0
    d_ref = d_dst.clone(); d_ref[:, 1:8] = d_src
    n = 4
    P, I = C.c_void_p * n, C.c_int64 * n
    dst = P(a.data_ptr(), b.data_ptr() + 4 * (3 * 50 + 1), c_dst.data_ptr(), d_dst.data_ptr() + 4)
    src = P(None, None, c_src.data_ptr() + 4 * 256, d_src.data_ptr())
    dp = I(4 * a.numel(), 4 * 50, 4 * 256, 4 * 9)
    sp = I(0, 0, 4 * 1032, 4 * 7)
    wd = I(4 * a.numel(), 4 * 41, 4 * 256, 4 * 7)
    rw = I(1, 28, 256, 5)
    assert lib.bg_fill_copy_batch(n, dst, dp, src, sp, wd, rw, st) == 0
    assert bool((a == 0).all()) and torch.equal(b, b_ref)
    assert torch.equal(c_dst, c_src[:, 256:512]) and torch.equal(d_dst, d_ref)
    assert lib.bg_fill_copy_batch(9, dst, dp, src, sp, wd, rw, st) == -1
    wd[1] = 4 * 41 + 2
    assert lib.bg_fill_copy_batch(n, dst, dp, src, sp, wd, rw, st) == -1
    assert lib.bg_fill_copy_batch(0, None, None, None, None, None, None, st) == 0


@pytest.mark.parametrize("H,nb", [(128, 11), (256, 3), (8, 2), (6, 3), (320, 2)])
def test_one_operator_on_several_matrices_in_one_launch(cuda, H, nb):
    """bg_spmm_batch_f32 (the story -> building mean of every block call's story term at once) against one bg_spmm_f32 per
    matrix: identical bits; widths outside the one-launch form (H % 4 != 0, H > 256) take the per-matrix path inside."""
    from building_gan_amd import _lib, ops
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g, p = _plan(_batch("dataset", 23, "jitter"), cuda)
    op = p.ops["story_building_mean"]
    c = op.csr
    S = c.n_rows
    gen = torch.Generator().manual_seed(H + nb)
    x = (torch.rand(nb, S, H, generator=gen) * 2 - 1).to(cuda)
    ref = torch.stack([ops.spmm_raw(op, x[b]) for b in range(nb)])
    out = torch.full((nb, S, H), float("nan"), device=cuda)
    rs = op.row_scale.data_ptr() if op.row_scale is not None else None
    cs = op.col_scale.data_ptr() if op.col_scale is not None else None
    assert lib.bg_spmm_batch_f32(x.data_ptr(), H, S * H, c.rowptr.data_ptr(), c.col.data_ptr(), rs, op.row_mode, cs, op.col_mode,
                                 out.data_ptr(), H, S * H, S, H, nb, st) == 0
    assert torch.equal(out, ref)
    assert lib.bg_spmm_batch_f32(x.data_ptr(), H, S * H, c.rowptr.data_ptr(), c.col.data_ptr(), rs, op.row_mode, cs, op.col_mode,
                                 out.data_ptr(), H, S * H, S, H, 0, st) == 0
    assert lib.bg_spmm_batch_f32(None, H, S * H, c.rowptr.data_ptr(), c.col.data_ptr(), rs, op.row_mode, cs, op.col_mode,
                                 out.data_ptr(), H, S * H, S, H, nb, st) == -1


@pytest.mark.parametrize("n,K,P,W,idx_n,mod", [(11520, 13, 3, 8, 34560, 11520), (5, 2, 3, 4, 1000, 7), (0, 1, 1, 4, 9, 4)])
def test_regroup_and_row_index_in_one_launch(cuda, n, K, P, W, idx_n, mod):
    """bg_regroup_iota_f32 against bg_regroup_f32 + bg_iota_mod_i32 (more index entries than regrouped chunks, and the
    other way round; no rows at all)."""
    from building_gan_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(n + K)
    src = torch.rand(max(n, 1) * K * P * W, generator=gen).to(cuda)
    for to_outer in (1, 0):
        want = torch.full_like(src, float("nan"))
        widx = torch.full((idx_n,), -1, dtype=torch.int32, device=cuda)
        if n:
            assert lib.bg_regroup_f32(src.data_ptr(), want.data_ptr(), n, K, P, W, to_outer, st) == 0
        assert lib.bg_iota_mod_i32(widx.data_ptr(), idx_n, mod, st) == 0
        got = torch.full_like(src, float("nan"))
        gidx = torch.full((idx_n,), -1, dtype=torch.int32, device=cuda)
        assert lib.bg_regroup_iota_f32(src.data_ptr(), got.data_ptr(), n, K, P, W, to_outer, gidx.data_ptr(), idx_n, mod, st) == 0
        assert torch.equal(gidx, widx) and torch.equal(gidx.cpu().long(), torch.arange(idx_n) % mod)
        if n:
            assert torch.equal(got, want)
    assert lib.bg_regroup_iota_f32(src.data_ptr(), got.data_ptr(), max(n, 1), K, P, W, 1, None, 5, mod, st) == -1


@pytest.mark.parametrize("n_rows,S,B,H", [(11520, 88, 8, 256), (301, 7, 2, 64), (40, 40, 1, 8)])
def test_pair_launchers_equal_their_two_single_calls(cuda, n_rows, S, B, H):
    """bg_decoder_fwd_pair_f32, bg_scatter_arg_pair_f32, bg_gather_arg_pair_f32 (two independent small launches of the
    critic's pooled head as one): bit for bit the two single calls, on rows of different counts and with -1 ids."""
    from building_gan_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(n_rows + H)
    rnd = lambda *s: (torch.rand(*s, generator=gen) * 2 - 1).to(cuda)
    nan = lambda *s: torch.full(s, float("nan"), device=cuda)
    # ---- decoders
    Hd = H // 2
    par = [[rnd(Hd, H), rnd(Hd), rnd(Hd), rnd(1)] for _ in range(2)]
    X = [rnd(2 * S, H), rnd(2 * B, H)]
    single = []
    for i in range(2):
        val, gp = nan(X[i].shape[0]), nan(H)
        w1, b1, w2, b2 = par[i]
        assert lib.bg_decoder_fwd_f32(X[i].shape[0], H, Hd, X[i].data_ptr(), H, w1.data_ptr(), H, b1.data_ptr(), w2.data_ptr(),
                                      b2.data_ptr(), val.data_ptr(), gp.data_ptr(), st) == 0
        single.append((val, gp))
    pv = [(nan(X[i].shape[0]), nan(H)) for i in range(2)]
    args = []
    for i in range(2):
        w1, b1, w2, b2 = par[i]
        args += [X[i].shape[0], X[i].data_ptr(), H, w1.data_ptr(), H, b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                 pv[i][0].data_ptr(), pv[i][1].data_ptr()]
    assert lib.bg_decoder_fwd_pair_f32(H, Hd, *args, st) == 0
    for i in range(2):
        assert torch.equal(pv[i][0], single[i][0]) and torch.equal(pv[i][1], single[i][1])
    assert lib.bg_decoder_fwd_pair_f32(H, Hd, *([0] + args[1:]), st) == -1            # no rows for the first decoder
    # ---- scatters: rows -> story (contiguous runs), rows -> building; the arg of a segment is one of its rows or -1
    seg_s = (torch.arange(n_rows) * S // n_rows).to(torch.int32)
    seg_b = (torch.arange(n_rows) * B // n_rows).to(torch.int32)
    seg_s[n_rows // 2] = -1                                               # a row without a segment

    def args_of(seg, n_seg):
        arg = torch.full((n_seg, H), -1, dtype=torch.int32)
        for q in range(n_seg):
            rows = torch.nonzero(seg == q).view(-1)
            if rows.numel():
                arg[q] = rows[torch.randint(0, rows.numel(), (H,), generator=gen)].to(torch.int32)
        arg[0, : H // 4] = -1
        return arg.to(cuda)
    arg_s, arg_b = args_of(seg_s, S), args_of(seg_b, B)
    seg_s, seg_b = seg_s.to(cuda), seg_b.to(cuda)
    g_s, g_b = rnd(S, H), rnd(1, H)                                       # (the building side: one row for every segment, pitch 0)
    one = [nan(n_rows, H), nan(n_rows, H)]
    assert lib.bg_scatter_arg_f32(g_b.data_ptr(), 0, arg_b.data_ptr(), seg_b.data_ptr(), one[0].data_ptr(), H, B, n_rows, H, st) == 0
    assert lib.bg_scatter_arg_f32(g_s.data_ptr(), H, arg_s.data_ptr(), seg_s.data_ptr(), one[1].data_ptr(), H, S, n_rows, H, st) == 0
    two = nan(2, n_rows, H)
    assert lib.bg_scatter_arg_pair_f32(g_b.data_ptr(), 0, arg_b.data_ptr(), seg_b.data_ptr(), two[0].data_ptr(), B,
                                       g_s.data_ptr(), H, arg_s.data_ptr(), seg_s.data_ptr(), two[1].data_ptr(), S,
                                       H, n_rows, H, st) == 0
    assert torch.equal(two[0], one[0]) and torch.equal(two[1], one[1])
    assert float(two[1][n_rows // 2].abs().max()) == 0.0 and int((two[0] != 0).sum()) > 0
    # (no segment ids: the two-call path inside the pair launcher)
    two.fill_(float("nan"))
    assert lib.bg_scatter_arg_pair_f32(g_b.data_ptr(), 0, arg_b.data_ptr(), None, two[0].data_ptr(), B,
                                       g_s.data_ptr(), H, arg_s.data_ptr(), None, two[1].data_ptr(), S, H, n_rows, H, st) == 0
    assert torch.equal(two[0], one[0]) and torch.equal(two[1], one[1])
    # ---- gathers
    x0, x1 = rnd(n_rows, H), rnd(n_rows, H + 4)[:, :H]                    # (a pitched source)
    ref = [nan(B, H), nan(S, H)]
    assert lib.bg_gather_arg_f32(x0.data_ptr(), H, arg_b.data_ptr(), ref[0].data_ptr(), H, B, H, st) == 0
    assert lib.bg_gather_arg_f32(x1.data_ptr(), H + 4, arg_s.data_ptr(), ref[1].data_ptr(), H, S, H, st) == 0
    got = [nan(B, H), nan(S, H)]
    assert lib.bg_gather_arg_pair_f32(x0.data_ptr(), H, arg_b.data_ptr(), got[0].data_ptr(), B,
                                      x1.data_ptr(), H + 4, arg_s.data_ptr(), got[1].data_ptr(), S, H, H, st) == 0
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert lib.bg_gather_arg_pair_f32(None, H, None, None, 0, x1.data_ptr(), H + 4, arg_s.data_ptr(), got[1].data_ptr(), S, H, H, st) == 0
    assert lib.bg_gather_arg_pair_f32(None, H, arg_b.data_ptr(), got[0].data_ptr(), B, x1.data_ptr(), H + 4, arg_s.data_ptr(),
                                      got[1].data_ptr(), S, H, H, st) == -1


@pytest.mark.parametrize("R,Rp,H2,Hd,with_add", [(16, 8, 256, 128, False), (176, 88, 256, 128, True), (5, 0, 40, 9, True)])
def test_decoder_backward_in_one_launch(cuda, R, Rp, H2, Hd, with_add):
    """bg_decoder_fwd_f32 / bg_decoder_bwd_f32 (a two-layer decoder without activation on a batch's pooled rows: the
    forward as one dot product per row with w2 W1, the backward + the penalty path's share as rank-one expressions)
    against the two Linears and autograd in float64."""
    from building_gan_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(R + Hd)
    rnd = lambda *s: (torch.rand(*s, generator=gen, dtype=torch.float64) * 2 - 1)
    X, w1, b1, w2, b2 = rnd(R, H2), rnd(Hd, H2).requires_grad_(), rnd(Hd).requires_grad_(), rnd(1, Hd).requires_grad_(), rnd(1).requires_grad_()
    Xr = X.clone().requires_grad_()
    dv, add = rnd(R), rnd(R, H2)
    P = rnd(max(Rp, 1), H2)[:Rp].clone().requires_grad_()          # the penalty's pooled rows: D is linear in them
    dgp = rnd(max(Rp, 1), H2)[:Rp]
    h = Xr @ w1.t() + b1
    val = (h @ w2.t() + b2).view(-1)
    pen = (((P @ w1.t() + b1) @ w2.t() + b2).view(-1)).sum() * 0         # (value unused: the penalty differentiates d D / d P)
    gP = torch.autograd.grad(((P @ w1.t() + b1) @ w2.t()).sum(), P, create_graph=True)[0] if Rp else None   # rows = w2 W1
    obj = (val * dv).sum() + ((gP * dgp).sum() if Rp else 0.0) + pen
    obj.backward()
    f = lambda t: t.detach().float().contiguous().to(cuda)
    gp = f((w2 @ w1).view(-1))
    out = {n: torch.full(s, float("nan"), device=cuda) for n, s in
           (("g_w1", (Hd, H2)), ("g_b1", (Hd,)), ("g_w2", (Hd,)), ("g_b2", (1,)), ("dX", (R, H2)))}
    Xc, hc, dvc, w1c, w2c, dgc, addc = f(X), f(h), f(dv), f(w1), f(w2.view(-1)), f(dgp) if Rp else None, f(add) if with_add else None
    b1c, b2c = f(b1), f(b2)
    val, gp2 = torch.full((R,), float("nan"), device=cuda), torch.full((H2,), float("nan"), device=cuda)
    st_ = torch.cuda.current_stream().cuda_stream
    assert lib.bg_decoder_fwd_f32(R, H2, Hd, Xc.data_ptr(), H2, w1c.data_ptr(), H2, b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(),
                                  val.data_ptr(), gp2.data_ptr(), st_) == 0
    vref = ((X @ w1.detach().t() + b1.detach()) @ w2.detach().t() + b2.detach()).view(-1)
    scale = float((X.abs() @ w1.detach().abs().t() @ w2.detach().abs().t()).max())
    assert float((val.cpu().double() - vref).abs().max()) < 2e-6 * scale
    assert float((gp2.cpu().double() - (w2.detach() @ w1.detach()).view(-1)).abs().max()) < 1e-5
    only_dx = torch.full((R, H2), float("nan"), device=cuda)               # frozen decoder: dX alone, no parameters read
    assert lib.bg_decoder_bwd_f32(R, 0, H2, Hd, dvc.data_ptr(), Xc.data_ptr(), H2, None, None, 0, None, 0, None, gp.data_ptr(),
                                  None, 0, None, 0, None, None, None, only_dx.data_ptr(), H2, st_) == 0
    assert float((only_dx.cpu().double() - Xr.grad).abs().max()) < 1e-5 * float(Xr.grad.abs().max())
    rc = lib.bg_decoder_bwd_f32(R, Rp, H2, Hd, dvc.data_ptr(), Xc.data_ptr(), H2, b1c.data_ptr(),
                                dgc.data_ptr() if Rp else None, H2, w1c.data_ptr(), H2, w2c.data_ptr(), gp.data_ptr(),
                                addc.data_ptr() if with_add else None, H2, out["g_w1"].data_ptr(), H2, out["g_b1"].data_ptr(),
                                out["g_w2"].data_ptr(), out["g_b2"].data_ptr(), out["dX"].data_ptr(), H2,
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = {"g_w1": w1.grad, "g_b1": b1.grad, "g_w2": w2.grad.view(-1), "g_b2": b2.grad,
           "dX": Xr.grad + (add if with_add else 0.0)}
    for n, r in ref.items():
        err = float((out[n].cpu().double() - r).abs().max()) / max(float(r.abs().max()), 1e-30)
        assert err < 1e-5, (n, err)              # fp32 sums of signed terms (sum dv cancels to a few percent of sum |dv|)
    assert lib.bg_decoder_bwd_f32(0, 0, H2, Hd, *([None] * 2), 0, None, None, 0, None, 0, None, None, None, 0, None, 0,
                                  None, None, None, None, 0, None) == -1
