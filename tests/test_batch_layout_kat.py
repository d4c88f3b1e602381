"""Known-answer test of the batch layout: graph.Batch.from_data_list against a 2-graph
batch whose every tensor was written out by hand from /root/reference/Data/VolumeDesignGraph.py:53-77, and one
ProgramBlock / VoxelBlock / Attention call of the oracle (CPU) and of the HIP networks (-m gpu) against float64
loops over those hand-written edge lists (tests/kat_two_graph.py)."""
import copy

import numpy as np
import pytest
import torch

import kat_two_graph as K


def _batch():
    from building_gan_amd.graph import Batch
    return Batch.from_data_list(K.graphs())


def test_from_data_list_matches_hand_written_batch():
    b = _batch()
    for key, want in K.EXPECT.items():
        got = getattr(b, key)
        want_t = torch.tensor(want, dtype=got.dtype)
        if key.endswith("_edge"):
            want_t = want_t.view(2, -1)
        assert got.shape == want_t.shape, key
        assert torch.equal(got, want_t), key
    for key, want in K.EXPECT_SLICES.items():
        assert list(b.__slices__[key]) == want, key
    assert b.num_graphs == 2 and b.data_id_str == ["000000", "000001"]
    assert np.array_equal(b.voxel_feature.numpy(), K.batched_voxel_feature())
    assert b.program_class_feature.shape == (K.NP, 6)
    assert b.program_class_feature.argmax(1).tolist() == [c % 6 for c in K.EXPECT["program_class_cluster"]]
    # the models' assumptions on the layout: dense story ids, voxels story-major, cross edges voxel-major
    assert sorted(set(b.voxel_floor_cluster.tolist())) == list(range(K.S))
    assert (b.cross_edge_voxel_index_select[1:] >= b.cross_edge_voxel_index_select[:-1]).all()


def test_detach_batch_keeps_layout():
    from building_gan_amd.graph import detach_batch
    b = _batch()
    d = detach_batch(b)
    for key in K.EXPECT:
        assert torch.equal(getattr(d, key), getattr(b, key)), key
    assert d.__slices__ == b.__slices__ and d.num_graphs == 2


def test_floor_level_literal():
    from oracle.networks import get_voxel_floor_level
    b = _batch()
    assert get_voxel_floor_level(b.voxel_floor_cluster, b.voxel_feature_batch).tolist() == K.FLOOR_LEVEL


def _inputs(H, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(K.NP, H, generator=g) * 2 - 1
    v = torch.rand(K.NV, H, generator=g) * 2 - 1
    em = torch.empty(K.NV, 2).exponential_(generator=g)
    ea = torch.empty(K.EC).exponential_(generator=g)
    return x, v, em, ea


def _check_attention(mask, att, new_v, want, tol):
    ms, mw, soft, arg, nv = want
    np.testing.assert_allclose(mask["soft"].detach().cpu().double().numpy().reshape(-1), ms, rtol=tol, atol=tol)
    hard = mask["hard"].detach().cpu().numpy().reshape(-1).round()
    assert hard.tolist() == [1.0 if w == 0 else 0.0 for w in mw]                 # column 0 of the one-hot
    np.testing.assert_allclose(att["soft"].detach().cpu().double().numpy().reshape(-1), soft, rtol=tol, atol=tol)
    hot = att["hard"].detach().cpu().numpy().reshape(-1).round()
    assert sorted(np.nonzero(hot)[0].tolist()) == sorted(arg.tolist())           # bit-exact pointer
    np.testing.assert_allclose(new_v.detach().cpu().double().numpy(), nv, rtol=tol, atol=tol)


def test_oracle_layers_match_float64_loops_over_the_literal_batch():
    from oracle import networks as om
    b = _batch()
    H = 8
    torch.manual_seed(1)
    pb, vb_g, vb_d, at = om.ProgramBlock(H), om.VoxelBlock(H, True), om.VoxelBlock(H, False), om.Attention(H)
    x, v, em, ea = _inputs(H)
    pos = b.voxel_feature[:, 3:6]
    got = pb(x, b.program_edge, b.program_class_cluster, b.program_target_ratio, b.program_class_feature_batch)
    np.testing.assert_allclose(got.detach().double().numpy(), K.program_block(pb, x.double().numpy()), rtol=1e-5, atol=1e-6)
    for blk in (vb_g, vb_d):
        got = blk(v, b.voxel_edge, b.voxel_feature_batch, pos, b.voxel_floor_cluster)
        want = K.voxel_block(blk, v.double().numpy(), pos.double().numpy())
        np.testing.assert_allclose(got.detach().double().numpy(), want, rtol=1e-5, atol=1e-6)
    mask, att, nv = at(x, v, b.cross_edge_program_index_select, b.cross_edge_voxel_index_select, (em, ea))
    _check_attention(mask, att, nv, K.attention(at, x.double().numpy(), v.double().numpy(), em.double().numpy(),
                                                ea.double().numpy()), 1e-5)


@pytest.mark.gpu
def test_hip_layers_match_float64_loops_over_the_literal_batch(cuda):
    """The HIP networks read the batch through graph.Batch and their own plans; the expectation never does."""
    from building_gan_amd import models as hm
    from building_gan_amd.plan import build_plan
    b = copy.deepcopy(_batch()).to(cuda)
    plan = build_plan(b)
    assert (plan.B, plan.S, plan.Np, plan.Nv, plan.Ec) == (K.NB, K.S, K.NP, K.NV, K.EC)
    assert plan.level.cpu().tolist() == K.FLOOR_LEVEL
    H = 128
    torch.manual_seed(1)
    pb, vb_g, vb_d, at = (hm.ProgramBlock(H).to(cuda), hm.VoxelBlock(H, True).to(cuda), hm.VoxelBlock(H, False).to(cuda),
                          hm.Attention(H).to(cuda))
    x, v, em, ea = _inputs(H)
    pos = K.batched_voxel_feature()[:, 3:6].astype(np.float64)
    pdm4 = hm.position_terms(b)
    xd, vd = x.to(cuda), v.to(cuda)
    with torch.no_grad():
        want = K.program_block(pb, x.double().numpy())
        got = pb(xd, plan, b.program_target_ratio)
        np.testing.assert_allclose(got.cpu().double().numpy(), want, rtol=1e-4, atol=1e-5)
        got = pb.forward_residual(xd, plan) - xd                       # the fused no-grad form
        np.testing.assert_allclose(got.cpu().double().numpy(), want, rtol=1e-4, atol=1e-5)
        for blk in (vb_g, vb_d):
            want = K.voxel_block(blk, v.double().numpy(), pos)
            got = blk(vd, plan, pdm4)                                  # literal two-Linear path
            np.testing.assert_allclose(got.cpu().double().numpy(), want, rtol=1e-4, atol=1e-5)
            got = blk.forward_composed(vd, plan, pdm4, blk.compose()) - vd     # collapsed weights, fused launches
            np.testing.assert_allclose(got.cpu().double().numpy(), want, rtol=1e-4, atol=2e-5)
        want = K.attention(at, x.double().numpy(), v.double().numpy(), em.double().numpy(), ea.double().numpy())
        mask, att, nv = at(xd, vd, plan, (em.to(cuda), ea.to(cuda)))
        _check_attention(mask, att, nv, want, 1e-4)
    # and with a graph (the autograd-node forms of the generator update)
    got = vb_g.forward_composed(vd.clone().requires_grad_(True), plan, pdm4, vb_g.compose()) - vd
    np.testing.assert_allclose(got.detach().cpu().double().numpy(), K.voxel_block(vb_g, v.double().numpy(), pos),
                               rtol=1e-4, atol=2e-5)
    mask, att, nv = at(xd, vd.clone().requires_grad_(True), plan, (em.to(cuda), ea.to(cuda)))
    _check_attention(mask, att, nv, want, 1e-4)
