"""Known-answer tests that pin the CPU oracle to the upstream semantics it restates.

The reference has no tests or golden vectors (parity unpinned), so every expected value below was
worked out by hand from the documented behaviour of torch_geometric 1.6.2 / torch_scatter 2.0.x.
"""
import math

import numpy as np
import torch

from oracle import plan as oplan
from oracle import pointer, scatter
from oracle.networks import Attention, MLP, get_voxel_floor_level, position_encoder


def test_message_passing_flow_source_to_target_mean():
    # edges (src -> dst): 0->1, 2->1, 1->0 ; node 2 has no in-edge
    ei = torch.tensor([[0, 2, 1], [1, 1, 0]])
    x = torch.tensor([[1.0], [10.0], [100.0]])
    seen = {}

    def msg(x_i, x_j):
        seen["i"], seen["j"] = x_i.clone(), x_j.clone()
        return x_j + 0.5 * x_i

    out = scatter.propagate_mean(msg, ei, 3, x=x)
    assert seen["j"].view(-1).tolist() == [1.0, 100.0, 10.0]      # *_j = source = edge_index[0]
    assert seen["i"].view(-1).tolist() == [10.0, 10.0, 1.0]       # *_i = target = edge_index[1]
    # node 1: mean(1+5, 100+5) = 55.5 ; node 0: 10+0.5 ; node 2 isolated -> 0 (sum / clamp(count,1))
    assert out.view(-1).tolist() == [10.5, 55.5, 0.0]


def test_scatter_max_empty_ties_and_gradient():
    src = torch.tensor([3.0, 7.0, 7.0, -2.0, -2.0], requires_grad=True)
    idx = torch.tensor([0, 0, 0, 2, 2])
    val, arg = scatter.scatter_max(src, idx, 4)
    assert val.tolist() == [7.0, 0.0, -2.0, 0.0]                  # empty rows -> 0
    assert arg.tolist() == [1, 5, 3, 5]                           # first index on ties; empty -> src.size(0)
    val.sum().backward()
    assert src.grad.tolist() == [0.0, 1.0, 0.0, 1.0, 0.0]         # gradient to the arg element only


def test_avg_pool_x_with_and_without_size():
    x = torch.tensor([[2.0], [4.0], [6.0], [8.0]])
    # with size: dim_size = (batch.max()+1)*size even when trailing clusters are empty
    out, b = scatter.avg_pool_x(torch.tensor([0, 0, 7, 7]), x, torch.tensor([0, 0, 1, 1]), size=6)
    assert b is None and out.shape == (12, 1)
    assert out.view(-1).tolist() == [3.0, 0, 0, 0, 0, 0, 0, 7.0, 0, 0, 0, 0]
    # without size: clusters relabelled by sorted unique, batch of a cluster = batch of a member
    out, b = scatter.avg_pool_x(torch.tensor([5, 9, 5, 9]), x, torch.tensor([0, 1, 0, 1]))
    assert out.view(-1).tolist() == [4.0, 6.0] and b.tolist() == [0, 1]
    mx, _ = scatter.max_pool_x(torch.tensor([5, 9, 5, 9]), x, torch.tensor([0, 1, 0, 1]))
    assert mx.view(-1).tolist() == [6.0, 8.0]


def test_global_max_pool_and_floor_level():
    x = torch.tensor([[1.0, -5.0], [3.0, -7.0], [2.0, -1.0]])
    assert scatter.global_max_pool(x, torch.tensor([0, 0, 1])).tolist() == [[3.0, -5.0], [2.0, -1.0]]
    vfc = torch.tensor([0, 0, 1, 2, 2, 3])          # dense global story ids; graph 1 starts at story 2
    vbi = torch.tensor([0, 0, 0, 1, 1, 1])
    assert get_voxel_floor_level(vfc, vbi).tolist() == [0, 0, 1, 0, 0, 1]
    assert oplan.floor_level(vfc.numpy(), vbi.numpy()).tolist() == [0, 0, 1, 0, 0, 1]


def test_glorot_bound_and_maybe_num_nodes():
    torch.manual_seed(0)
    att = Attention(128)
    a = math.sqrt(6.0 / 129.0)
    assert float(att.theta.abs().max()) <= a and float(att.theta.abs().max()) > 0.5 * a
    assert scatter.maybe_num_nodes(torch.tensor([0, 4, 2])) == 5


def test_dense_gumbel_softmax_defaults_and_straight_through():
    logits = torch.tensor([[0.0, math.log(3.0)]])
    noise = torch.ones(1, 2)                         # -log(1) = 0 gumbel -> plain softmax, tau = 1
    y = pointer.dense_gumbel_softmax(logits, noise)
    np.testing.assert_allclose(y.numpy(), [[0.25, 0.75]], rtol=1e-6)
    soft = torch.tensor([[0.25, 0.75]], requires_grad=True)
    hard = pointer.straight_through_one_hot(soft, -1)
    assert hard.detach().tolist() == [[0.0, 1.0]]
    (hard * torch.tensor([[2.0, 5.0]])).sum().backward()
    assert soft.grad.tolist() == [[2.0, 5.0]]        # gradient of the soft value passes straight through


def test_segment_gumbel_softmax_tau_is_minus_one():
    # zero gumbel noise (exp noise = 1): y = -src, so the pointer picks the arg-MIN of src
    src = torch.tensor([1.0, 3.0, 2.0, 0.5, 0.5])
    idx = torch.tensor([0, 0, 0, 1, 1])
    soft, hard = pointer.segment_gumbel_softmax(src, idx, exp_noise=torch.ones(5))
    e = np.exp(-np.array([1.0, 3.0, 2.0]) + 1.0)
    np.testing.assert_allclose(soft[:3].numpy(), e / e.sum(), rtol=1e-6)
    np.testing.assert_allclose(soft[3:].numpy(), [0.5, 0.5], rtol=1e-6)
    assert hard.round().tolist() == [1.0, 0.0, 0.0, 1.0, 0.0]     # min of src; tie -> first edge
    # straight-through value is (1 - s) + s, not necessarily exactly 1; off-winners are exactly 0
    assert hard[1].item() == 0.0 and hard[2].item() == 0.0 and hard[4].item() == 0.0


def test_find_max_out_program_index_masked_voxel_takes_first_edge():
    cev = torch.tensor([0, 0, 1, 1])
    cep = torch.tensor([4, 5, 4, 5])
    logit = torch.tensor([[0.0], [1.0], [0.0], [0.0]])           # voxel 1 is masked: all-zero logits
    assert pointer.find_max_out_program_index(logit, cev, cep, 2).view(-1).tolist() == [5, 4]


def test_position_table_and_mlp_layout():
    pe = position_encoder(128, 20)
    assert pe.shape == (20, 128) and pe[0, 0].item() == 0.0 and pe[0, 1].item() == 1.0
    np.testing.assert_allclose(pe[3, 2].item(), math.sin(3 * math.exp(-2 * math.log(10000.0) / 128)), rtol=1e-6)
    seq = MLP([4, 3, 2], act="leaky")
    assert [type(m).__name__ for m in seq] == ["Linear", "LeakyReLU", "Linear", "LeakyReLU"]
    assert [type(m).__name__ for m in MLP([4, 3, 2])] == ["Linear", "Linear"]


def test_mean_aggregation_equals_dense_adjacency_fp64():
    rng = np.random.default_rng(0)
    n, e, h = 30, 120, 5
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    x = rng.standard_normal((n, h))
    A = np.zeros((n, n))
    np.add.at(A, (dst, src), 1.0)
    deg = np.maximum(A.sum(1, keepdims=True), 1.0)
    ref = (A / deg) @ x
    out = scatter.scatter_mean(torch.from_numpy(x)[torch.from_numpy(src)], torch.from_numpy(dst), n)
    np.testing.assert_allclose(out.numpy(), ref, rtol=1e-12, atol=1e-12)
    rp, col, perm = oplan.csr_by_key(dst, src, n)
    for i in range(n):                                # CSR rows keep COO order
        assert perm[rp[i]:rp[i + 1]].tolist() == sorted(perm[rp[i]:rp[i + 1]].tolist())
        assert np.array_equal(col[rp[i]:rp[i + 1]], src[dst == i])


def test_program_ratio_hand_case():
    class G:
        pass
    g = G()
    g.voxel_feature = torch.zeros(3, 7)
    g.voxel_feature[:, 6] = torch.tensor([2.0, 3.0, 5.0])
    g.cross_edge_voxel_index_select = torch.tensor([0, 0, 1, 1, 2, 2])
    g.cross_edge_program_index_select = torch.tensor([0, 1, 0, 1, 0, 1])
    g.program_class_feature = torch.zeros(2, 6)
    g.program_class_cluster = torch.tensor([0, 3])
    g.program_target_ratio = torch.zeros(6)
    g.program_target_ratio_batch = torch.zeros(6, dtype=torch.long)
    g.FAR = torch.zeros(1)
    att = torch.tensor([[1.0], [0.0], [0.0], [1.0], [1.0], [0.0]])
    mask = torch.tensor([[1.0], [1.0], [0.0]])       # voxel 2 masked out
    ratio, pw, far = pointer.get_program_ratio(g, att, mask, 6)
    assert pw.view(-1).tolist() == [2.0, 3.0] and far.view(-1).tolist() == [5.0]
    np.testing.assert_allclose(ratio.view(-1).numpy(), [0.4, 0, 0, 0.6, 0, 0], rtol=1e-6)
