"""A hand-built 2-graph batch -- TEST INFRASTRUCTURE.

Every tensor of the batched layout below was written out BY HAND from the batching rules of
/root/reference/Data/VolumeDesignGraph.py:53-77 (per-key increments and concatenation dimension) and PyG's
``follow_batch`` vectors (/root/reference/train.py:43), not produced by building-gan_amd.graph.Batch; and the
expected layer outputs are computed by plain float64 loops over those literal edge lists (one edge / node /
segment at a time, following /root/reference/Model/models.py:52-73, 104-125, 144-172), sharing no code with the
oracle's scatter helpers or with the HIP plans.  So a wrong offset rule in graph.Batch, a wrong CSR in the plan or
a wrong reading of PyG's flow direction each show up as a mismatch.

Graph A: 2 stories; voxels v0 v1 (story 0), v2 (story 1); program nodes p0 p1 (story 0), p2 (story 1).
Graph B: 3 stories; voxels v0 (story 0), v1 v2 (story 1), v3 (story 2); program nodes p0 | p1 p2 | p3.
"""
import numpy as np
import torch

# ---------------------------------------------------------------- the two graphs, each in its own numbering
A = dict(
    data_id_str="000000",
    program_floor_cluster=[0, 0, 1], story_level_feature=[0.0, 0.0, 1.0], program_class_cluster=[2, 0, 2],
    program_edge=[[0, 1, 1, 2], [1, 0, 2, 1]],
    neg_same_story_program_edge=[[], []], neg_diff_story_program_edge=[[0, 2], [2, 0]],
    program_target_ratio=[0.25, 0.0, 0.75, 0.0, 0.0, 0.0], FAR=[2.0],
    voxel_floor_cluster=[0, 0, 1], voxel_projection_cluster=[0, 1, 0],
    voxel_edge=[[0, 1, 0, 2], [1, 0, 2, 0]], voxel_edge_mask=[1.0, 1.0, 0.0, 0.0],
    cross_edge_program_index_select=[0, 1, 0, 1, 2], cross_edge_voxel_index_select=[0, 0, 1, 1, 2],
    voxel_label_program_node_index=[0, -1, 2], voxel_label_class=[2, -1, 2],
    voxel_xyz=[[0, 0, 0], [1, 0, 0], [0, 0, 1]], voxel_weight=[1.0, 0.5, 0.75],
)
B = dict(
    data_id_str="000001",
    program_floor_cluster=[0, 1, 1, 2], story_level_feature=[0.0, 1.0, 1.0, 2.0], program_class_cluster=[1, 4, 1, 5],
    program_edge=[[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]],
    neg_same_story_program_edge=[[], []], neg_diff_story_program_edge=[[0, 2, 0, 3, 1, 3], [2, 0, 3, 0, 3, 1]],
    program_target_ratio=[0.0, 0.5, 0.0, 0.0, 0.125, 0.375], FAR=[3.5],
    voxel_floor_cluster=[0, 1, 1, 2], voxel_projection_cluster=[0, 0, 1, 0],
    voxel_edge=[[0, 1, 1, 2, 1, 3], [1, 0, 2, 1, 3, 1]], voxel_edge_mask=[0.0, 0.0, 1.0, 1.0, 0.0, 0.0],
    cross_edge_program_index_select=[0, 1, 2, 1, 2, 3], cross_edge_voxel_index_select=[0, 1, 1, 2, 2, 3],
    voxel_label_program_node_index=[0, 2, -1, 3], voxel_label_class=[1, 1, -1, 5],
    voxel_xyz=[[0, 0, 0], [0, 0, 1], [1, 0, 1], [0, 0, 2]], voxel_weight=[0.9, 0.6, 0.8, 1.0],
)

# ---------------------------------------------------------------- the batch [A, B], written out by hand
# increments of B's entries (VolumeDesignGraph.py:60-72): program ids +3 (A's program nodes), voxel ids +3
# (A's voxels), story ids +2 (A's max story level + 1), class slots +6, projection clusters +2 (A's max + 1).
# voxel_label_program_node_index takes the program increment like any other entry: B's void marker -1 becomes 2.
EXPECT = dict(
    program_edge=[[0, 1, 1, 2, 3, 4, 4, 5, 5, 6], [1, 0, 2, 1, 4, 3, 5, 4, 6, 5]],                 # cat along dim -1
    neg_same_story_program_edge=[[], []],
    neg_diff_story_program_edge=[[0, 2, 3, 5, 3, 6, 4, 6], [2, 0, 5, 3, 6, 3, 6, 4]],
    voxel_edge=[[0, 1, 0, 2, 3, 4, 4, 5, 4, 6], [1, 0, 2, 0, 4, 3, 5, 4, 6, 4]],
    cross_edge_program_index_select=[0, 1, 0, 1, 2, 3, 4, 5, 4, 5, 6],
    cross_edge_voxel_index_select=[0, 0, 1, 1, 2, 3, 4, 4, 5, 5, 6],
    voxel_label_program_node_index=[0, -1, 2, 3, 5, 2, 6],
    program_floor_cluster=[0, 0, 1, 2, 3, 3, 4],
    voxel_floor_cluster=[0, 0, 1, 2, 3, 3, 4],
    program_class_cluster=[2, 0, 2, 7, 10, 7, 11],
    voxel_projection_cluster=[0, 1, 0, 2, 2, 3, 2],
    story_level_feature=[0.0, 0.0, 1.0, 0.0, 1.0, 1.0, 2.0],                                        # no increment
    program_target_ratio=[0.25, 0.0, 0.75, 0.0, 0.0, 0.0, 0.0, 0.5, 0.0, 0.0, 0.125, 0.375],
    FAR=[2.0, 3.5],
    voxel_edge_mask=[1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0],
    program_class_feature_batch=[0, 0, 0, 1, 1, 1, 1],
    voxel_feature_batch=[0, 0, 0, 1, 1, 1, 1],
    program_target_ratio_batch=[0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1],
)
EXPECT_SLICES = dict(program_edge=[0, 4, 10], voxel_edge=[0, 4, 10], voxel_feature=[0, 3, 7], program_class_feature=[0, 3, 7],
                     cross_edge_voxel_index_select=[0, 5, 11], program_target_ratio=[0, 6, 12], FAR=[0, 1, 2],
                     neg_same_story_program_edge=[0, 0, 0], neg_diff_story_program_edge=[0, 2, 8])
NP, NV, EC, S, NB = 7, 7, 11, 5, 2
FLOOR_LEVEL = [0, 0, 1, 0, 1, 1, 2]          # story id minus the first story id of the voxel's own graph


def _voxel_feature(g):
    """[dimension / 100 (3) | coordinate / 100 (3) | weight]  (/root/reference/Data/GraphConstructor.py:127)."""
    xyz = np.asarray(g["voxel_xyz"], dtype=np.float64) * np.array([4.0, 3.0, 3.0])
    dims = np.tile(np.array([4.0, 3.0, 3.0]), (len(xyz), 1))
    return np.concatenate([dims / 100.0, xyz / 100.0, np.asarray(g["voxel_weight"])[:, None]], 1).astype(np.float32)


def _label(g):
    out = np.zeros((len(g["voxel_label_class"]), 6), dtype=np.float32)
    for i, c in enumerate(g["voxel_label_class"]):
        if c >= 0:
            out[i, c] = 1.0
    return out


def make_graph(g):
    from building_gan_amd.graph import VolumeDesignGraph
    L = lambda v: torch.tensor(v, dtype=torch.long)
    E = lambda v: torch.tensor(v, dtype=torch.long).view(2, -1)
    F_ = lambda v: torch.tensor(v, dtype=torch.float32)
    onehot = np.eye(6, dtype=np.float32)[np.asarray(g["program_class_cluster"])]
    return VolumeDesignGraph(
        data_id_str=g["data_id_str"], program_floor_cluster=L(g["program_floor_cluster"]),
        story_level_feature=F_(g["story_level_feature"]), program_class_feature=torch.from_numpy(onehot),
        program_edge=E(g["program_edge"]), neg_same_story_program_edge=E(g["neg_same_story_program_edge"]),
        neg_diff_story_program_edge=E(g["neg_diff_story_program_edge"]),
        program_class_cluster=L(g["program_class_cluster"]), program_target_ratio=F_(g["program_target_ratio"]),
        FAR=F_(g["FAR"]), voxel_floor_cluster=L(g["voxel_floor_cluster"]),
        voxel_projection_cluster=L(g["voxel_projection_cluster"]), voxel_feature=torch.from_numpy(_voxel_feature(g)),
        voxel_bool=F_([float(c >= 0) for c in g["voxel_label_class"]]), voxel_label=torch.from_numpy(_label(g)),
        voxel_edge=E(g["voxel_edge"]), voxel_edge_mask=F_(g["voxel_edge_mask"]),
        cross_edge_program_index_select=L(g["cross_edge_program_index_select"]),
        cross_edge_voxel_index_select=L(g["cross_edge_voxel_index_select"]),
        voxel_label_program_node_index=L(g["voxel_label_program_node_index"]))


def graphs():
    return [make_graph(A), make_graph(B)]


def batched_voxel_feature():
    return np.concatenate([_voxel_feature(A), _voxel_feature(B)], 0)


# ---------------------------------------------------------------- float64 loops over the literal batch
def _lin(mod):
    return mod.weight.detach().double().cpu().numpy(), mod.bias.detach().double().cpu().numpy()


def _leaky(x, slope=0.01):
    return np.where(x > 0, x, slope * x)


def program_block(block, x):
    """models.py:52-73: message = Wm [x_i; x_j] + bm per edge j -> i (i = edge[1]), mean over the in-edges of i
    (0 without in-edges); class-slot means (slot = graph * 6 + class) times the slot's target ratio, taken back
    to the nodes; update = LeakyReLU(Wu [x; agg; pooled] + bu)."""
    Wm, bm = _lin(block.messageMLP[0])
    Wu, bu = _lin(block.updateMLP[0])
    src, dst = EXPECT["program_edge"]
    n, H = x.shape
    agg, cnt = np.zeros((n, H)), np.zeros(n)
    for j, i in zip(src, dst):
        agg[i] += Wm @ np.concatenate([x[i], x[j]]) + bm
        cnt[i] += 1
    agg /= np.maximum(cnt, 1)[:, None]
    slots = EXPECT["program_class_cluster"]
    pooled = np.zeros((n, H))
    for i in range(n):
        members = [k for k in range(n) if slots[k] == slots[i]]
        pooled[i] = np.mean([x[k] for k in members], 0) * EXPECT["program_target_ratio"][slots[i]]
    return np.stack([_leaky(Wu @ np.concatenate([x[i], agg[i], pooled[i]]) + bu) for i in range(n)])


def voxel_block(block, v, pos):
    """models.py:144-172 (generator flavour): message = Wm [v_i; v_j; pos_i - pos_j] + bm, mean over in-edges;
    story mean; building mean = mean of the building's story means; update = LeakyReLU(Wu [v; agg; story;
    building - story] + bu).  Discriminator flavour (no story terms, no activation) when the block says so."""
    Wm, bm = _lin(block.messageMLP[0])
    Wu, bu = _lin(block.updateMLP[0])
    src, dst = EXPECT["voxel_edge"]
    n, H = v.shape
    agg, cnt = np.zeros((n, H)), np.zeros(n)
    for j, i in zip(src, dst):
        agg[i] += Wm @ np.concatenate([v[i], v[j], pos[i] - pos[j]]) + bm
        cnt[i] += 1
    agg /= np.maximum(cnt, 1)[:, None]
    if not block.if_agg_story_in_update:
        return np.stack([Wu @ np.concatenate([v[i], agg[i]]) + bu for i in range(n)])
    story_of, graph_of = EXPECT["voxel_floor_cluster"], EXPECT["voxel_feature_batch"]
    story_mean = {s: np.mean([v[i] for i in range(n) if story_of[i] == s], 0) for s in set(story_of)}
    story_graph = {story_of[i]: graph_of[i] for i in range(n)}
    bld_mean = {g: np.mean([story_mean[s] for s in story_mean if story_graph[s] == g], 0) for g in set(graph_of)}
    return np.stack([_leaky(Wu @ np.concatenate([v[i], agg[i], story_mean[story_of[i]],
                                                 bld_mean[graph_of[i]] - story_mean[story_of[i]]]) + bu)
                     for i in range(n)])


def attention(att, x, v, exp_mask, exp_att):
    """models.py:104-125 + util.py:13-28: two-way Gumbel-softmax mask (tau 1), cross-edge logits
    theta . tanh(NN_v v_i + NN_p x_j), segment Gumbel-softmax with tau = -1 over each voxel's cross edges,
    v' = v + mask_soft * sum_e soft_e x[cep_e].  Returns (mask_soft, mask_hard_winner, soft_att, arg edge of each
    voxel, v')."""
    W1, b1 = _lin(att.dec[0])
    W2, b2 = _lin(att.dec[1])
    Wv, bv = _lin(att.NN_v)
    Wp, bp = _lin(att.NN_p)
    theta = att.theta.detach().double().cpu().numpy().reshape(-1)
    cep, cev = EXPECT["cross_edge_program_index_select"], EXPECT["cross_edge_voxel_index_select"]
    n = v.shape[0]
    mask_soft, mask_win = np.zeros(n), np.zeros(n, dtype=np.int64)
    for i in range(n):
        y = W2 @ (W1 @ v[i] + b1) + b2 - np.log(exp_mask[i])
        e = np.exp(y - y.max())
        p = e / e.sum()
        mask_soft[i], mask_win[i] = p[0], int(np.argmax(p))
    logit = np.array([theta @ np.tanh(Wv @ v[i] + bv + Wp @ x[j] + bp) for j, i in zip(cep, cev)])
    y = (logit - np.log(exp_att)) / -1.0
    soft = np.zeros(len(cep))
    arg = np.zeros(n, dtype=np.int64)
    for i in range(n):
        es = [e for e in range(len(cep)) if cev[e] == i]
        m = max(y[e] for e in es)
        ex = {e: np.exp(y[e] - m) for e in es}
        tot = sum(ex.values()) + 1e-16
        for e in es:
            soft[e] = ex[e] / tot
        arg[i] = max(es, key=lambda e: (soft[e], -e))          # first edge on ties
    new_v = v.copy()
    for e, (j, i) in enumerate(zip(cep, cev)):
        new_v[i] += mask_soft[i] * soft[e] * x[j]
    return mask_soft, mask_win, soft, arg, new_v


# ---------------------------------------------------------------- round 3: the rest of the path, by the same loops
# Whole networks on the literal batch: the generator's call schedule (/root/reference/Model/models.py:185-203,
# 241-250), its output stage (:128-141 with /root/reference/util_graph.py:44-48), the discriminator with both max
# pools (:215-229, 260-264), the area pipeline (/root/reference/util_graph.py:21-41) and the gradient penalty's VALUE
# by central differences (/root/reference/Model/losses.py:73-89).  float64 throughout, one node / edge / segment at a
# time, no oracle code, no plan.
def position_table(d_model, max_len=20):
    """models.py:13-19."""
    pe = np.zeros((max_len, d_model))
    for p in range(max_len):
        for k in range(0, d_model, 2):
            f = np.exp(k * (-np.log(10000.0) / d_model))
            pe[p, k] = np.sin(p * f)
            pe[p, k + 1] = np.cos(p * f)
    return pe


def _split_feature():
    f = batched_voxel_feature().astype(np.float64)
    return f[:, 3:6], np.concatenate([f[:, 0:3], f[:, 6:]], 1)          # models.py:45-49: (pos, the rest)


def program_gnn(gnn, z):
    """models.py:83-92: enc([class one-hot | story level | z]) then layer_num residual applications of ONE block."""
    W, b = _lin(gnn.enc[0])
    onehot = np.eye(6)[[c % 6 for c in EXPECT["program_class_cluster"]]]
    x = np.stack([W @ np.concatenate([onehot[i], [EXPECT["story_level_feature"][i]], z[i]]) + b for i in range(NP)])
    for _ in range(gnn.layer_num):
        x = x + program_block(gnn.block, x)
    return x


def generator_forward(G, program_z, noises):
    """models.py:241-250 with VoxelGNN_G.forward (:185-203): attention once before the layers, then after layer i for
    even i > 0; the mask / att of the LAST call go to the output stage.  ``noises`` = [(Exp(1) [Nv, 2], Exp(1) [Ec])]
    in call order.  Returns (out, soft_out, mask_soft, mask_win, att_soft, arg edge per voxel, program index per
    voxel, calls made)."""
    x = program_gnn(G.programGNN, program_z)
    gnn = G.voxelGNN
    pos, rest = _split_feature()
    W, b = _lin(gnn.enc[0])
    pe = position_table(W.shape[0])
    v = np.stack([W @ rest[i] + b + pe[FLOOR_LEVEL[i]] for i in range(NV)])
    call = 0
    _, _, _, _, v = attention(gnn.attention, x, v, noises[call][0], noises[call][1])
    call += 1
    last = None
    for i in range(gnn.layer_num):
        v = v + voxel_block(gnn.block, v, pos)
        if i % 2 == 0 and i != 0:
            ms, mw, soft, arg, v = attention(gnn.attention, x, v, noises[call][0], noises[call][1])
            call += 1
            last = (ms, mw, soft, arg)
    ms, mw, soft, arg = last
    out, soft_out, idx = construct_output(ms, mw, soft, arg)
    return out, soft_out, ms, mw, soft, arg, idx, call


def straight_through(hard, soft):
    """util.py:10 / :26: (hard - soft) + soft, in this order."""
    return (hard - soft) + soft


def construct_output(mask_soft, mask_win, att_soft, arg):
    """models.py:128-141: per voxel, the program node of the cross edge with the largest att_hard * mask_hard (first
    edge on ties: a masked voxel's logits are all zero), and the class one-hots of the voxel's edges summed with the
    hard / soft attention, times the hard mask."""
    cep, cev = EXPECT["cross_edge_program_index_select"], EXPECT["cross_edge_voxel_index_select"]
    onehot = np.eye(6)[[c % 6 for c in EXPECT["program_class_cluster"]]]
    mask_hard = np.array([straight_through(1.0 if mask_win[i] == 0 else 0.0, mask_soft[i]) for i in range(NV)])
    att_hard = np.array([straight_through(1.0 if arg[cev[e]] == e else 0.0, att_soft[e]) for e in range(EC)])
    out, soft_out, idx = np.zeros((NV, 6)), np.zeros((NV, 6)), np.zeros(NV, dtype=np.int64)
    for i in range(NV):
        es = [e for e in range(EC) if cev[e] == i]
        best = es[0]
        for e in es[1:]:
            if att_hard[e] * mask_hard[i] > att_hard[best] * mask_hard[i]:
                best = e
        idx[i] = cep[best]
        for e in es:
            out[i] += att_hard[e] * onehot[cep[e]]
            soft_out[i] += att_soft[e] * onehot[cep[e]]
        out[i] *= mask_hard[i]
        soft_out[i] *= mask_hard[i]
    return out, soft_out, idx


def segment_max(v, seg_of, n_seg):
    """torch_scatter.scatter_max as global_max_pool uses it (models.py:262-263): per segment and column the largest
    value and the FIRST row that holds it; a segment without rows gives 0 and no row."""
    val, arg = np.zeros((n_seg, v.shape[1])), -np.ones((n_seg, v.shape[1]), dtype=np.int64)
    for s in range(n_seg):
        rows = [i for i in range(v.shape[0]) if seg_of[i] == s]
        for h in range(v.shape[1]):
            for i in rows:
                if arg[s, h] < 0 or v[i, h] > val[s, h]:
                    val[s, h], arg[s, h] = v[i, h], i
    return val, arg


def _mlp2(seq, x):
    (W1, b1), (W2, b2) = _lin(seq[0]), _lin(seq[1])
    return W2 @ (W1 @ x + b1) + b2                       # act = "": no activation (train.py:65 for WGAN-GP)


def discriminator_forward(D, label):
    """models.py:215-229 + 260-264: [feature_enc(rest) + pe[level] | label_enc(label)], layer_num residual D-flavour
    blocks, max over each story / each graph, the two decoders.  Returns (building [B], story [S])."""
    gnn = D.voxelGNN_D
    pos, rest = _split_feature()
    Wf, bf = _lin(gnn.feature_enc[0])
    Wl, bl = _lin(gnn.label_enc[0])
    pe = position_table(Wf.shape[0])
    v = np.stack([np.concatenate([Wf @ rest[i] + bf + pe[FLOOR_LEVEL[i]], Wl @ label[i] + bl]) for i in range(NV)])
    for _ in range(gnn.layer_num):
        v = v + voxel_block(gnn.block, v, pos)
    story, _ = segment_max(v, EXPECT["voxel_floor_cluster"], S)
    graph, _ = segment_max(v, EXPECT["voxel_feature_batch"], NB)
    return (np.array([_mlp2(D.building_dec, graph[g])[0] for g in range(NB)]),
            np.array([_mlp2(D.story_dec, story[s])[0] for s in range(S)]))


def program_ratio(att, mask, area_col=6):
    """util_graph.py:21-41: masked voxel area -> cross edge -> program node -> (graph, class) slot -> graph."""
    cep, cev = EXPECT["cross_edge_program_index_select"], EXPECT["cross_edge_voxel_index_select"]
    area = batched_voxel_feature().astype(np.float64)[:, area_col]
    slots, slot_graph = EXPECT["program_class_cluster"], EXPECT["program_target_ratio_batch"]
    program_weight = np.zeros(NP)
    for e in range(EC):
        program_weight[cep[e]] += att[e] * (mask[cev[e]] * area[cev[e]])
    class_weight = np.zeros(len(slot_graph))
    for p in range(NP):
        class_weight[slots[p]] += program_weight[p]
    far = np.zeros(NB)
    for s, g in enumerate(slot_graph):
        far[g] += class_weight[s]
    ratio = np.array([class_weight[s] / (far[slot_graph[s]] + 1e-16) for s in range(len(slot_graph))])
    return ratio, program_weight, far


def gp_sample(label, out, u):
    """losses.py:76-79: hard one-hot of the interpolate (straight-through VALUES), zero where the interpolate is."""
    sample = np.zeros_like(label)
    for i in range(NV):
        mixed = label[i] * u[i] + out[i] * (1.0 - u[i])
        hard = np.zeros(6)
        hard[int(np.argmax(mixed))] = 1.0                 # torch.max: first index on ties
        sample[i] = straight_through(hard, mixed) * (1.0 if mixed.max() != 0 else 0.0)
    return sample


def gradient_penalty(D, label, out, u, h=1e-6):
    """losses.py:81-87 with the two gradients taken by central differences of the summed outputs (the discriminator
    is piecewise linear in its label input, so the differences are exact up to float64 rounding away from a switch of
    a max): mean over voxels of (|d sum(D_b) / d sample_i|_2 - 1)^2, the same for the story output, and their sum."""
    sample = gp_sample(label, out, u)
    gb, gs = np.zeros_like(sample), np.zeros_like(sample)
    for i in range(NV):
        for c in range(6):
            up, dn = sample.copy(), sample.copy()
            up[i, c] += h
            dn[i, c] -= h
            (bu, su), (bd, sd) = discriminator_forward(D, up), discriminator_forward(D, dn)
            gb[i, c] = (bu.sum() - bd.sum()) / (2 * h)
            gs[i, c] = (su.sum() - sd.sum()) / (2 * h)
    gp_b = np.mean((np.sqrt((gb ** 2).sum(1)) - 1.0) ** 2)
    gp_s = np.mean((np.sqrt((gs ** 2).sum(1)) - 1.0) ** 2)
    return gp_b + gp_s, gp_b, gp_s, sample
