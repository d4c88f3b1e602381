"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank holds half of the graphs; with the
Trainer's global-count loss scaling the SUM all-reduce of the flat gradient buffers must equal the
gradients of one process holding the whole batch.  The Trainer's host logic is
exercised with the oracle networks injected (the HIP networks need a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=5):
    from building_gan_amd.synthetic import make_graph
    rng = np.random.default_rng(seed)
    shapes = [(2, 3, 2, [2, 3]), (1, 2, 2, [1]), (3, 2, 3, [3, 1, 2]), (2, 2, 2, [2, 2])]
    return [make_graph(rng, f, x, y, pps, data_id=i) for i, (f, x, y, pps) in enumerate(shapes)]


def _nets():
    from oracle.networks import DiscriminatorVoxel, Generator
    torch.manual_seed(77)
    return Generator(7, 7, 32, 8, 1, 3), DiscriminatorVoxel(7, 6, 32, 3)


def _inputs(full, n_att=2):
    from building_gan_amd.synthetic import batch_sizes
    sz = batch_sizes(full)
    g = torch.Generator().manual_seed(99)
    pz = torch.rand(sz["Np"], 8, generator=g)
    vz = torch.rand(sz["Nv"], 8, generator=g)
    noise = [(torch.empty(sz["Nv"], 2).exponential_(generator=g), torch.empty(sz["Ec"]).exponential_(generator=g))
             for _ in range(n_att)]
    u = torch.rand(sz["Nv"], 1, generator=g)
    return pz, vz, noise, u


def _slice(full, lo, hi, pz, vz, noise, u):
    sl = full.__slices__
    p0, p1 = sl["program_class_feature"][lo], sl["program_class_feature"][hi]
    v0, v1 = sl["voxel_feature"][lo], sl["voxel_feature"][hi]
    c0, c1 = sl["cross_edge_voxel_index_select"][lo], sl["cross_edge_voxel_index_select"][hi]
    return pz[p0:p1], vz[v0:v1], [(a[v0:v1], b[c0:c1]) for a, b in noise], u[v0:v1]


def _trainer():
    from building_gan_amd.train import Trainer
    G, D = _nets()
    tr = Trainer(torch.device("cpu"), noise_dim=8, generator=G, discriminator=D, tr_weight=1.0, far_weight=0.5)
    from oracle.pointer import get_program_ratio
    tr.program_ratio_fn = get_program_ratio      # the HIP pipeline needs a GPU plan
    return tr


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from building_gan_amd.graph import Batch
        from building_gan_amd.train import global_counts
        torch.set_num_threads(1)
        data = _make()
        full = Batch.from_data_list(data)
        pz, vz, noise, u = _inputs(full)
        per = len(data) // world
        mine = Batch.from_data_list(data[rank * per:(rank + 1) * per])
        a = _slice(full, rank * per, (rank + 1) * per, pz, vz, noise, u)
        tr = _trainer()
        # the counts fetched ahead of the iteration (train.prefetch_counts, what Trainer.prefetch / fit do for the next
        # batch) are what the direct all-reduce gives, and are consumed once
        from building_gan_amd.train import prefetch_counts
        direct = global_counts(mine)
        prefetch_counts(mine)
        kept = mine.__dict__["_bg_counts"]          # (world size, the summed counts [host tensor], event or None)
        assert kept[0] == world and tuple(int(x) for x in kept[1].tolist()) == direct and kept[2] is None
        counts = global_counts(mine)
        assert counts == direct and "_bg_counts" not in mine.__dict__
        gen, stats = tr._phase_d(mine, counts, True, *a)
        local = tr.opt_d.grad.clone()
        tr.opt_d.all_reduce()
        d_grad = tr.opt_d.grad.clone()
        # the one-shot exchange (all-gather, then a local sum in rank order) gives the same sums
        tr.opt_d.grad.copy_(local)
        tr.opt_d.all_reduce(exchange="gather_sum")
        assert torch.equal(tr.opt_d.grad, d_grad) and tr.opt_d._gathered.shape == (world, local.numel())
        assert torch.equal(tr.opt_d._gathered[rank], local)
        tr._phase_g(mine, counts, gen)
        tr.opt_g.all_reduce()
        g_grad = tr.opt_g.grad.clone()
        d_loss = stats["d_loss"].clone()
        dist.all_reduce(d_loss)
        if rank == 0:
            ret["counts"] = counts
            ret["d_grad"], ret["g_grad"], ret["d_loss"] = d_grad.numpy(), g_grad.numpy(), float(d_loss)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process_union_batch():
    from building_gan_amd.graph import Batch
    from building_gan_amd.synthetic import batch_sizes
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    # single process, whole batch
    data = _make()
    full = Batch.from_data_list(data)
    sz = batch_sizes(full)
    pz, vz, noise, u = _inputs(full)
    tr = _trainer()
    counts = (sz["B"], sz["S"], sz["Nv"])
    assert tuple(ret["counts"]) == counts
    gen, stats = tr._phase_d(full, counts, True, pz, vz, noise, u)
    d_grad = tr.opt_d.grad.clone().numpy()
    tr._phase_g(full, counts, gen)
    g_grad = tr.opt_g.grad.clone().numpy()
    np.testing.assert_allclose(ret["d_loss"], float(stats["d_loss"]), rtol=1e-5)
    np.testing.assert_allclose(ret["d_grad"], d_grad, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(ret["g_grad"], g_grad, rtol=2e-4, atol=1e-6)
    assert np.abs(d_grad).max() > 0 and np.abs(g_grad).max() > 0


def test_flat_buffers_alias_parameters_and_gradients():
    from building_gan_amd.train import FlatAdam
    G, _ = _nets()
    opt = FlatAdam(G)
    p = next(G.parameters())
    assert p.data_ptr() == opt.flat.data_ptr() and p.grad.data_ptr() == opt.grad.data_ptr()
    (p.sum() * 2).backward()
    assert float(opt.grad[: p.numel()].min()) == 2.0       # autograd accumulated in place into the flat buffer
    opt.zero_grad()
    assert float(opt.grad.abs().max()) == 0.0 and p.grad.data_ptr() == opt.grad.data_ptr()
