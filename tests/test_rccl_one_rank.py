"""The data-parallel call path on the hardware at hand: ONE rank, backend "nccl" (= RCCL on ROCm).

north_star asks for an RCCL gradient all-reduce for both networks; the reference has no collective at all
(/root/reference/util_graph.py:149-158 is marked "Not used", /root/reference/train.py:22-23 picks cuda:0).  With one
rank every collective is the identity, so the trainer normally skips them; ``train.FORCE_COLLECTIVES`` issues them
anyway.  These tests run the SAME iterations with and without the collectives and require bit-identical losses,
parameters, gradients and optimizer state: what RCCL adds is calls, not arithmetic.  Covered: broadcast_parameters,
FlatAdam.all_reduce for D (every iteration) and G (every 5th), global_counts, Trainer.prefetch with the counts'
collective + read-back on the plan's side stream, the group-wide collapse guard (all-reduce MAX), and the captured
iteration cut into graph segments around the eager collectives (train_iteration_graphed).

No N > 1 measurement exists on this pool (one GPU per call): the 8-GPU curve is the driver's.
"""
import copy
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def rccl_group(cuda):
    import torch.distributed as dist
    from building_gan_amd import train
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=cuda)
    calls = {"all_reduce": 0, "broadcast": 0}
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)

    def bc(*a, **k):
        calls["broadcast"] += 1
        return real_bc(*a, **k)
    dist.all_reduce, dist.broadcast = ar, bc
    try:
        yield dist.group.WORLD, calls
    finally:
        dist.all_reduce, dist.broadcast = real_ar, real_bc
        train.FORCE_COLLECTIVES = False
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _batches(cuda):
    from building_gan_amd.synthetic import make_batch
    return [make_batch("small", seed=s, ragged="jitter").to(cuda) for s in (21, 22, 23)]


def _snapshot(tr, stats):
    torch.cuda.synchronize()
    return ({k: v.detach().clone() for k, v in stats.items() if torch.is_tensor(v)},
            tr.opt_g.flat.clone(), tr.opt_d.flat.clone(), tr.opt_d.grad.clone(), tr.opt_g.grad.clone(),
            tr.opt_d.m.clone(), tr.opt_d.v.clone(), int(tr.opt_d.t.item()), int(tr.opt_g.t.item()))


def _assert_same(a, b, what):
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), (what, "loss", k, a[0][k], b[0][k])
    for i, name in enumerate(("G parameters", "D parameters", "D gradient buffer", "G gradient buffer", "D first moment",
                              "D second moment"), start=1):
        assert torch.equal(a[i], b[i]), (what, name, (a[i] - b[i]).abs().max())
    assert a[7:] == b[7:], (what, "step counts", a[7:], b[7:])


def _eager_run(cuda, group, force, batches, exchange="all_reduce"):
    from building_gan_amd import train
    train.FORCE_COLLECTIVES = force
    torch.manual_seed(0)
    tr = train.Trainer(cuda, group=group, collapse_check_every=3)
    tr.grad_exchange = exchange
    tr.broadcast_parameters(0)
    torch.manual_seed(77)
    snaps = []
    for b in batches:
        for key in ("_bg_plan", "_bg_plan_pending", "_bg_counts"):
            b.__dict__.pop(key, None)
    for i in range(6):                                  # i = 0 and 5 carry the generator update, i = 0 and 3 the guard
        b, nxt = batches[i % len(batches)], batches[(i + 1) % len(batches)]
        for key in ("_bg_plan", "_bg_plan_pending", "_bg_counts"):
            nxt.__dict__.pop(key, None)
        tr.prefetch(nxt)                                # plan + (with collectives) the counts on the side stream
        if force:
            assert "_bg_counts" in nxt.__dict__, "prefetch_counts did not run its collective"
        stats = tr.train_iteration(b, i)
        snaps.append(_snapshot(tr, stats))
    return snaps


def test_eager_iterations_with_rccl_collectives_equal_iterations_without(cuda, rccl_group):
    import torch.distributed as dist
    group, calls = rccl_group
    assert dist.get_backend(group) == "nccl" and dist.get_world_size(group) == 1
    batches = _batches(cuda)
    plain = _eager_run(cuda, group, False, batches)
    assert calls == {"all_reduce": 0, "broadcast": 0}, calls       # one rank, not forced: nothing is issued
    forced = _eager_run(cuda, group, True, batches)
    # 6 D all-reduces + 2 G all-reduces + 6 prefetched counts + the first batch's counts (nobody prefetched them) +
    # 2 guard maxima; 2 broadcasts (G and D parameters)
    assert calls["broadcast"] == 2 and calls["all_reduce"] == 6 + 2 + 6 + 1 + 2, calls
    for i, (a, b) in enumerate(zip(plain, forced)):
        _assert_same(a, b, "eager iteration %d" % i)
    # the one-shot gradient exchange (RCCL all-gather + local sum in rank order) on the same path: the 8 gradient
    # all-reduces become all-gathers, everything else as before, same bits
    n0 = dict(calls)
    gathered = _eager_run(cuda, group, True, batches, exchange="gather_sum")
    assert calls["all_reduce"] - n0["all_reduce"] == 6 + 1 + 2, (n0, calls)
    for i, (a, b) in enumerate(zip(plain, gathered)):
        _assert_same(a, b, "eager iteration %d, gather_sum" % i)


def _graphed_run(cuda, group, force, batch):
    from building_gan_amd import train
    train.FORCE_COLLECTIVES = force
    torch.manual_seed(0)
    tr = train.Trainer(cuda, group=group, collapse_check_every=0)
    tr.broadcast_parameters(0)
    torch.manual_seed(78)
    tr.capture(batch)
    snaps = []
    for i in range(6):
        stats = tr.train_iteration_graphed(i)
        snaps.append(_snapshot(tr, stats))
    return snaps


def test_graph_replays_with_rccl_collectives_between_segments_equal_replays_without(cuda, rccl_group):
    """capture() cuts the iteration at the collectives; the all-reduces stay eager RCCL calls between graph launches."""
    group, calls = rccl_group
    batch = _batches(cuda)[0]
    plain = _graphed_run(cuda, group, False, copy.copy(batch))
    n0 = dict(calls)
    forced = _graphed_run(cuda, group, True, copy.copy(batch))
    # replays: 6 D all-reduces + 2 G all-reduces (capture's eager warm-up iterations and counts add their own)
    assert calls["all_reduce"] - n0["all_reduce"] >= 8 and calls["broadcast"] - n0["broadcast"] == 2, (n0, calls)
    for i, (a, b) in enumerate(zip(plain, forced)):
        _assert_same(a, b, "graph replay %d" % i)
