"""plan.LinOp.pair: an operator and its transpose without a reference cycle (host logic, no GPU).

A plan builds ~25 such pairs per batch.  While the two held each other, a dropped plan -- its Csr's, ~55 tensor views and
with them the plan's device memory -- waited for the cyclic collector, whose full collections showed up as 50-70 ms stalls
of the training loop (tools/lab/mode_probe.py)."""
import gc
import weakref

import torch


def _pair():
    from building_gan_amd.plan import Csr, LinOp
    z = lambda n: torch.zeros(n, dtype=torch.int32)
    fwd, rev = Csr(z(3), z(2), None, 2, 2, 2), Csr(z(3), z(2), None, 2, 2, 2)
    return LinOp.pair(fwd, rev, torch.ones(2), 2), fwd, rev


def test_pair_is_freed_by_reference_counting_alone():
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        a, fwd, rev = _pair()
        b = a.T
        assert b.T is a and a.T is b and b.csr is rev and (b.col_mode, b.row_mode) == (2, 0) and b.col_scale is a.row_scale
        refs = [weakref.ref(a), weakref.ref(b), weakref.ref(fwd), weakref.ref(rev)]
        del a, b, fwd, rev
        assert all(r() is None for r in refs)
    finally:
        if was:
            gc.enable()


def test_transpose_kept_alone_finds_or_rebuilds_the_first_operator():
    a, fwd, rev = _pair()
    b = a.T
    scale = a.row_scale
    del a
    a2 = b.T                                   # the first operator is gone: the same operator again, from the Csr b keeps
    assert a2.csr is fwd and a2.row_scale is scale and a2.row_mode == 2 and a2.col_scale is None and a2.T is b
    assert b.T is a2                           # .. and found again while somebody holds it
    from building_gan_amd.plan import LinOp
    assert LinOp(fwd).T is None                # an operator made without a transpose has none
