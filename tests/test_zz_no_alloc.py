"""bg_segmax_static_f32 (the entry point that allocates nothing: chunk partials in the library's static device
buffer) against bg_segmax_f32 with a caller-provided workspace, through the C ABI: values and arg-max indices
bit-exact, on two calls in a row, the second with a smaller input (the buffer keeps the first call's contents),
and the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pool(lib, dev, n_rows, seg_len, H, R, seed, static):
    from building_gan_amd._lib import check, ptr, stream
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(0, seg_len + 1, (n_rows,), generator=g)
    lens[0] = seg_len                                   # at least one long segment, some empty ones
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(lens, 0).to(torch.int32)
    n = int(rowptr[-1])
    x = (torch.rand(n, H, generator=g) * 2 - 1)
    x[1] = x[0]                                         # a tie inside segment 0: the first row must win
    col = torch.randperm(n, generator=g).to(torch.int32)
    rowptr, col, x = rowptr.to(dev), col.to(dev), x.to(dev)
    chunk_ptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    check(lib.bg_plan_chunks(ptr(rowptr), n_rows, R, ptr(chunk_ptr), stream()), "bg_plan_chunks")
    ub = n // R + n_rows
    out = torch.empty(n_rows, H, device=dev)
    arg = torch.empty(n_rows, H, dtype=torch.int32, device=dev)
    if static:
        rc = lib.bg_segmax_static_f32(ptr(x), H, ptr(rowptr), ptr(col), ptr(chunk_ptr), R, ub, ptr(out), H, ptr(arg),
                                      n_rows, H, stream())
    else:
        pv = torch.empty(ub, H, device=dev)
        pa = torch.empty(ub, H, dtype=torch.int32, device=dev)
        rc = lib.bg_segmax_f32(ptr(x), H, ptr(rowptr), ptr(col), ptr(chunk_ptr), R, ub, ptr(pv), ptr(pa), ptr(out), H,
                               ptr(arg), n_rows, H, stream())
    torch.cuda.synchronize()
    return rc, out, arg, (x, col, rowptr)


def test_static_scratch_segmax_equals_workspace_segmax(cuda):
    from building_gan_amd import _lib
    lib = _lib.load()
    for n_rows, seg_len, H, R, seed in ((37, 900, 256, 64, 1), (5, 130, 128, 64, 2)):     # the second call is smaller
        rc_s, out_s, arg_s, (x, col, rowptr) = _pool(lib, cuda, n_rows, seg_len, H, R, seed, True)
        rc_w, out_w, arg_w, _ = _pool(lib, cuda, n_rows, seg_len, H, R, seed, False)
        assert rc_s == 0 and rc_w == 0
        assert torch.equal(out_s, out_w) and torch.equal(arg_s, arg_w)
        # and against a plain per-segment maximum (first row on ties, 0 / -1 for an empty segment)
        rp = rowptr.cpu().tolist()
        xc, cc = x.cpu(), col.cpu().long()
        for r in (0, 1, n_rows - 1):
            rows = cc[rp[r]:rp[r + 1]]
            if len(rows) == 0:
                assert float(out_s[r].abs().max()) == 0.0 and int(arg_s[r].max()) == -1
                continue
            vals, pos = xc[rows].max(0)
            assert torch.equal(out_s[r].cpu(), vals)
            first = torch.stack([rows[(xc[rows][:, h] == vals[h]).nonzero()[0, 0]] for h in range(0, H, 37)])
            assert torch.equal(arg_s[r].cpu()[::37].long(), first)


def test_static_scratch_limit_is_an_error(cuda):
    from building_gan_amd import _lib
    from building_gan_amd._lib import ptr, stream
    lib = _lib.load()
    d = torch.zeros(8, dtype=torch.int32, device=cuda)
    f = torch.zeros(8, device=cuda)
    rc = lib.bg_segmax_static_f32(ptr(f), 256, ptr(d), None, ptr(d), 64, (1 << 20) // 256 + 1, ptr(f), 256, ptr(d),
                                  1, 256, stream())
    assert rc == -5 and b"static scratch" in lib.bg_error_string(rc)
    assert lib.bg_segmax_static_f32(ptr(f), 256, ptr(d), None, None, 64, 1, ptr(f), 256, ptr(d), 1, 256, stream()) == -1
