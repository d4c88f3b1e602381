"""Generator / DiscriminatorVoxel with the reference's module interface, parameter names and
outputs (/root/reference/Model/models.py:232-264), computed by the HIP kernels of this package.

Differences in *formulation* (not in result, within fp32 rounding; DESIGN.md "refactors"):
  * messageMLP has no activation (models.py:56,149), so the mean over in-edges of
    W[x_i; x_j; dpos] + b is computed per NODE as W[x_i; mean_j x_j; mean dpos] + b on nodes
    with at least one in-edge: one gather-mean + one node-level GEMM instead of an edge-level
    GEMM and a scatter (an E/N-fold cut in flops and no [E, 2H+3] intermediate);
  * NN_v / NN_p of the attention (models.py:113-116) are applied per node, then gathered;
  * torch.cat feeding a Linear is fused into the GEMM's K loop.
Gumbel / Exp(1) noise can be passed in (``noise``: one (exp_mask[Nv,2], exp_att[Ec]) pair per
Attention call) so that runs are reproducible against the CPU oracle.
"""
from __future__ import annotations

import contextlib
import math
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ops import ACT_LEAKY, ACT_NONE, SpMM, SpMMW, linear
from .plan import LinOp, build_plan, iota_i32

NUMBER_OF_CLASS = 6  # /root/reference/Data/GraphConstructor.py:9
COMPOSE_STACK = True   # run the discriminator's 12 composed layers as one autograd node (ops.AffineStack)
FUSE_GATHER_STORY_MAX_ROWS = 1 << 16   # gather-mean + story projection share a launch up to this many voxels
FUSE_ATTENTION_NODE = True   # Attention call with a graph: one autograd node with a hand-written backward
FUSE_GBLOCK = True      # generator block with a graph: one autograd node with a hand-written backward
SHARE_WEIGHT_GRADS = True   # the calls of one module in a forward sum their weight gradients in the producing launches
FUSE_COEF_INPUTS = True   # collapsed discriminator: the recurrence's parameter rows by one launch (ops.PolyCoefD)
FUSE_GENERATOR_FORWARD = True   # graph-free Generator.forward as ONE host call (bg_generator_forward_f32)
FUSE_PSTACK = True        # generator update: ProgramGNN as ONE autograd node, one host call each way (ops.ProgramStack)
FUSE_GSTACK = True        # generator update: VoxelGNN_G below the program GNN as ONE autograd node, one host call each way (ops.VoxelStackG)
FUSE_COMPOSE_D = True     # collapsed discriminator: the block's composition inside the coefficient node (ops.PolyCoefDW)
FUSE_CRITIC_HEAD = True   # WGAN-GP critic update below the coefficient table as ONE host call each way (ops.CriticHead)
COLLAPSE_D = True      # evaluate the discriminator's (linear) voxel GNN as a polynomial in the aggregation operator


def position_encoder(d_model: int, max_len: int = 20) -> torch.Tensor:
    """Sinusoid table, /root/reference/Model/models.py:13-19."""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * freq)
    pe[:, 1::2] = torch.cos(pos * freq)
    return pe


def MLP(dims, act: str = "") -> nn.Sequential:
    """Same container layout as the reference (models.py:29-42) so state_dict keys agree."""
    mods = []
    for a, b in zip(dims[:-1], dims[1:]):
        mods.append(nn.Linear(a, b, bias=True))
        if "leaky" in act:
            mods.append(nn.LeakyReLU(negative_slope=0.01))
        elif "relu" in act:
            mods.append(nn.ReLU())
        elif "tanh" in act:
            mods.append(nn.Tanh())
        elif "sigmoid" in act:
            mods.append(nn.Sigmoid())
    return nn.Sequential(*mods)


def run_mlp(seq: nn.Sequential, xs, row_scale=None, pad_to: int = 0):
    """Evaluate an MLP container on the HIP GEMM; ``xs`` may be a list (fused concatenation)."""
    mods = list(seq)
    i = 0
    x = xs
    while i < len(mods):
        lin = mods[i]
        W = lin.weight
        if pad_to and W.shape[1] < pad_to:
            W = F.pad(W, (0, pad_to - W.shape[1]))
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(nxt, nn.LeakyReLU):
            x = linear(x, W, lin.bias, ACT_LEAKY, nxt.negative_slope, row_scale)
            i += 2
        else:
            x = linear(x, W, lin.bias, ACT_NONE, 0.0, row_scale)
            i += 1
            if nxt is not None and not isinstance(nxt, nn.Linear):
                x = nxt(x)
                i += 1
        row_scale, pad_to = None, 0
    return x


def extract_pos(feature):
    """(pos [N, 3] view, the other columns [N, F - 3] copied together).  For a data tensor (no gradient) the copy is kept
    on the tensor object: every pass of every iteration over a batch asks for the same split of the same features."""
    if feature.requires_grad:
        return feature[:, 3:6], torch.cat((feature[:, 0:3], feature[:, 6:]), dim=-1)
    kept = getattr(feature, "_bg_split", None)
    if kept is None or kept[0] != feature._version:
        kept = (feature._version, torch.cat((feature[:, 0:3], feature[:, 6:]), dim=-1))
        feature._bg_split = kept
    return feature[:, 3:6], kept[1]


PDM4_PAD = 1.0   # the 4th (pad) column of the position terms.  Every weight that multiplies it is zero-padded there, so
                 # products do not see it; a weight GRADIENT against it is the column sum of the output gradient,
                 # i.e. the bias gradient, which ops.GBlock reads there instead of launching a column sum.


def _pos_terms(plan, pos):
    """Per-node mean of (pos_i - pos_j) over in-edges, padded to 4 columns with PDM4_PAD (cached per plan)."""
    cached = plan.ops.get("pdm4")
    if cached is None:
        cached = ops.pos_terms_raw(plan.vox_mean, pos.detach(), PDM4_PAD)   # one launch, straight off the feature columns
        plan.ops["pdm4"] = cached
    return cached


def position_terms(graph):
    """Compute (and keep on the plan) the per-voxel mean position differences of a batch."""
    return _pos_terms(build_plan(graph), extract_pos(graph.voxel_feature)[0])


def _level_op(plan, max_len: int) -> LinOp:
    key = ("level_op", max_len)
    op = plan.ops.get(key)
    if op is None:
        from .plan import Csr
        n = plan.Nv
        if plan.ops["max_level"] >= max_len:
            raise IndexError(f"a building has {plan.ops['max_level'] + 1} stories but the position table has "
                             f"{max_len} rows (reference models.py:13,192)")
        op = LinOp(Csr(iota_i32(n + 1, plan.device), plan.level, None, n, max_len, n))
        plan.ops[key] = op
    return op


class ProgramBlock(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.aggr = "mean"
        self.messageMLP = MLP([2 * hidden_dim, hidden_dim])
        self.updateMLP = MLP([3 * hidden_dim, hidden_dim], act="leaky")

    def forward_residual(self, feature, plan):
        """feature + block(feature) when nothing is differentiated: both aggregations in one launch
        (plan.prog_pair), the residual in the second product's epilogue."""
        n = feature.shape[0]
        both = ops.spmm_raw(plan.prog_pair, feature)                 # [2 Np, H]: in-edge mean | ratio x class-slot mean
        aggr = run_mlp(self.messageMLP, [feature, both[:n]], row_scale=plan.prog_has)
        lin, act = self.updateMLP[0], self.updateMLP[1]
        return linear([feature, aggr, both[n:]], lin.weight, lin.bias, ACT_LEAKY, act.negative_slope, residual=feature)

    def forward(self, feature, plan, class_feature):
        gm = SpMM.apply(feature, plan.prog_mean)
        aggr = run_mlp(self.messageMLP, [feature, gm], row_scale=plan.prog_has)
        pooled = SpMM.apply(feature, plan.slot_mean) * class_feature.view(-1, 1)
        on_node = SpMM.apply(pooled, plan.slot_gather)
        return run_mlp(self.updateMLP, [feature, aggr, on_node])


class ProgramGNN(nn.Module):
    def __init__(self, input_dim, hidden_dim, noise_dim, layer_num):
        super().__init__()
        self.layer_num = layer_num
        self.enc = MLP([input_dim + noise_dim, hidden_dim])
        self.block = ProgramBlock(hidden_dim)

    def forward(self, cf, slf, tr, z, plan):
        program_in = torch.cat([cf, slf.view(-1, 1), z], dim=-1)
        if (FUSE_PSTACK and torch.is_grad_enabled() and program_in.is_cuda and "prog_pair" in plan.ops
                and "story_building_mean" in plan.ops and len(self.enc) == 1 and program_in.dtype == torch.float32
                and self.enc[0].weight.shape[0] % 4 == 0 and plan.Np > 0 and plan.Ep > 0):
            # a graph is being built (generator update): the whole program GNN as one autograd node
            _generator_plan_struct(plan)
            blk = self.block
            return ops.ProgramStack.apply(self.enc[0].weight, self.enc[0].bias, blk.messageMLP[0].weight, blk.messageMLP[0].bias,
                                          blk.updateMLP[0].weight, blk.updateMLP[0].bias, program_in, plan, self.layer_num)
        x = run_mlp(self.enc, program_in)
        fused = not torch.is_grad_enabled() and "prog_pair" in plan.ops
        for _ in range(self.layer_num):
            x = self.block.forward_residual(x, plan) if fused else x + self.block(x, plan, tr)
        return x


class Attention(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.dec = MLP([hidden_dim, hidden_dim // 2, 2])
        self.NN_v = nn.Linear(hidden_dim, hidden_dim)
        self.NN_p = nn.Linear(hidden_dim, hidden_dim)
        self.theta = nn.Parameter(torch.empty(1, hidden_dim))
        a = math.sqrt(6.0 / (1 + hidden_dim))          # torch_geometric inits.glorot
        nn.init.uniform_(self.theta, -a, a)

    def shared_terms(self, program_graph_feature, stacked=None):
        """What the calls of one generator forward have in common (the module and the program features are
        the same in all of them, reference models.py:194,201):
          * NN_p(program features), evaluated once instead of once per call;
          * ONE weight for everything that reads the voxel features: the mask decoder has no activation
            (models.py:98), so its two Linears are the [2, H] map W2 W1 with bias W2 b1 + b2, stacked under
            NN_v's [H, H] (rows padded to a multiple of 4).  A call then runs one GEMM instead of three."""
        b = linear(program_graph_feature, self.NN_p.weight, self.NN_p.bias)
        Wcat, bcat = stacked if stacked is not None else self.stacked_weights()
        return b, Wcat, bcat

    def stacked_weights(self):
        """(Wcat [H + 4, H], bcat): NN_v with the composed mask decoder stacked under it -- a function of the
        parameters alone (None, None when the decoder is not two plain Linears)."""
        mods = [m for m in self.dec if isinstance(m, nn.Linear)]
        if len(mods) != len(self.dec) or len(mods) != 2:
            return None, None
        W1, b1, W2, b2 = mods[0].weight, mods[0].bias, mods[1].weight, mods[1].bias
        Wd = ops.MatMul.apply(W2, W1, False, False)                                    # [2, H]
        bd = linear(b1.view(1, -1), W2, b2).view(-1)                                   # W2 b1 + b2 (the library's Linear)
        Wcat = torch.cat([self.NN_v.weight, Wd, ops.zeros_view(2, W1.shape[1], device=Wd.device)], dim=0)  # [H + 4, H]
        bcat = torch.cat([self.NN_v.bias, bd, ops.zeros_view(2, device=bd.device)])
        return Wcat, bcat

    def forward(self, program_graph_feature, voxel_feature, plan, noise=None, shared=None, grads=None):
        """``grads`` = (ops.SharedGrads, is-first-call) when the calls of one forward sum their weight gradients."""
        exp_mask, exp_att = (None, None) if noise is None else noise
        b, Wcat, bcat = shared if shared is not None else self.shared_terms(program_graph_feature)
        H = self.NN_v.weight.shape[0]
        if (Wcat is not None and not torch.is_grad_enabled() and 0 < plan.cross_v.max_len <= 64
                and H % 64 == 0):
            # nothing will be differentiated: NN_v as one [Nv,H] x [H,H] product (no ragged 4-column tile), then
            # ONE launch for the rest of the call, the two-row composed mask decoder included
            a = linear(voxel_feature, Wcat[:H], bcat[:H])
            if exp_mask is None:
                exp_mask = torch.empty(voxel_feature.shape[0], 2, dtype=a.dtype, device=a.device).exponential_()
            if exp_att is None:
                exp_att = torch.empty(plan.Ec, dtype=a.dtype, device=a.device).exponential_()
            soft0, hard0, soft_att, hard_att, new_voxel_feature = ops.attention_forward_raw(
                a, None, b, program_graph_feature, voxel_feature, self.theta, plan, exp_mask, exp_att, -1.0,
                wd=Wcat[H:H + 2], bd=bcat[H:H + 2])
            return ({"hard": hard0, "soft": soft0}, {"hard": hard_att.view(-1, 1), "soft": soft_att.view(-1, 1)},
                    new_voxel_feature)
        if (Wcat is not None and FUSE_ATTENTION_NODE and torch.is_grad_enabled() and 0 < plan.cross_v.max_len <= 64
                and H % 64 == 0 and H % 4 == 0):
            # a graph is being built (generator update): the whole call as one autograd node
            if exp_mask is None:
                exp_mask = torch.empty(voxel_feature.shape[0], 2, dtype=torch.float32, device=voxel_feature.device).exponential_()
            if exp_att is None:
                exp_att = torch.empty(plan.Ec, dtype=torch.float32, device=voxel_feature.device).exponential_()
            soft0, hard0, soft_att, hard_att, new_voxel_feature = ops.AttentionNode.apply(
                voxel_feature, program_graph_feature, b, Wcat, bcat, self.theta, exp_mask, exp_att, plan, -1.0,
                *(grads if grads is not None else (None, True)))
            return ({"hard": hard0, "soft": soft0}, {"hard": hard_att.view(-1, 1), "soft": soft_att.view(-1, 1)},
                    new_voxel_feature)
        if Wcat is not None:
            a, logits, _ = linear(voxel_feature, Wcat, bcat).split([H, 2, 2], dim=1)
        else:
            logits = run_mlp(self.dec, voxel_feature)
            a = linear(voxel_feature, self.NN_v.weight, self.NN_v.bias)
        if exp_mask is None:
            exp_mask = torch.empty(logits.shape, dtype=logits.dtype, device=logits.device).exponential_()
        if not torch.is_grad_enabled() and 0 < plan.cross_v.max_len <= 64:
            # nothing will be differentiated: the rest of the call is one launch
            if exp_att is None:
                exp_att = torch.empty(plan.Ec, dtype=logits.dtype, device=logits.device).exponential_()
            soft0, hard0, soft_att, hard_att, new_voxel_feature = ops.attention_forward_raw(
                a, logits, b, program_graph_feature, voxel_feature, self.theta, plan, exp_mask, exp_att, -1.0)
            return ({"hard": hard0, "soft": soft0}, {"hard": hard_att.view(-1, 1), "soft": soft_att.view(-1, 1)},
                    new_voxel_feature)
        soft0, hard0 = ops.MaskGumbel.apply(logits, exp_mask)
        out_mask = {"hard": hard0, "soft": soft0}
        att = ops.CrossLogits.apply(a, b, self.theta, plan)
        if exp_att is None:
            exp_att = torch.empty_like(att).exponential_()
        soft_att, hard_att, _ = ops.SegGumbelSoftmax.apply(att, exp_att, plan, -1.0)
        out_att = {"hard": hard_att.view(-1, 1), "soft": soft_att.view(-1, 1)}

        gathered = SpMMW.apply(program_graph_feature, soft_att, plan.cross_v_sum)
        new_voxel_feature = torch.addcmul(voxel_feature, out_mask["soft"], gathered)
        return out_mask, out_att, new_voxel_feature

    @staticmethod
    def construct_output(program_class_feature, num_voxel, att, mask, plan, cross_edge_program_index):
        if (not torch.is_grad_enabled() and program_class_feature.shape[1] <= 8
                and plan.ops["flags"]["cross_by_voxel"][1] == 0):
            # nothing will be differentiated: one launch (slots are in edge order on voxel-major cross edges)
            return ops.construct_output_raw(mask["hard"], att["hard"], att["soft"], plan, program_class_feature)
        on_edge = SpMM.apply(mask["hard"], plan.edge_voxel_gather)
        arg = ops.seg_argmax(att["hard"] * on_edge, plan)
        max_out_program_index = cross_edge_program_index[arg.long()]
        hard_label = mask["hard"] * SpMMW.apply(program_class_feature, att["hard"].view(-1), plan.cross_v_sum)
        soft_label = mask["hard"] * SpMMW.apply(program_class_feature, att["soft"].view(-1), plan.cross_v_sum)
        return hard_label, soft_label, max_out_program_index.view(-1)


class VoxelBlock(nn.Module):
    def __init__(self, hidden_dim, if_agg_story_in_update=True):
        super().__init__()
        self.aggr = "mean"
        self.if_agg_story_in_update = if_agg_story_in_update
        self.messageMLP = MLP([2 * hidden_dim + 3, hidden_dim])
        self.updateMLP = (MLP([4 * hidden_dim, hidden_dim], act="leaky") if if_agg_story_in_update
                          else MLP([2 * hidden_dim, hidden_dim]))

    def compose(self):
        """messageMLP has no activation, so on graphs where every voxel has an in-edge the block's
        two Linears collapse into one:  Wu[v; Wm[v; Av; p] + bm; ...] = Wc[v; Av; p] + bc (+ story terms)
        with Wc = [Wu1 + Wu2 Wm1 | Wu2 Wm2 | Wu2 Wm3], bc = Wu2 bm + bu.  Halves (D) / thirds (G) the
        flops of every pass; the small weight products are differentiable GEMMs, so gradients reach
        messageMLP / updateMLP through the chain rule.  Returns (Wc [H,2H+4], bc, Wsb [H,2H]) with
        Wsb = [Wu3 - Wu4 | Wu4] acting on [story mean | building mean]."""
        Wm, bm = self.messageMLP[0].weight, self.messageMLP[0].bias
        Wu, bu = self.updateMLP[0].weight, self.updateMLP[0].bias
        H = Wm.shape[0]
        # [H, 2H+8]: [Wm | 0 | bm | 0 0 0] -- the pad keeps the row pitch a multiple of 16 bytes (vector loads)
        Wm_ext = torch.cat([Wm, ops.zeros_view(H, 1, device=Wm.device), bm.view(-1, 1),
                            ops.zeros_view(H, 3, device=Wm.device)], dim=1)
        # split / one cat per result: each is ONE autograd node (a slice would cost a zero-fill + add in backward)
        parts = Wu.split(H, dim=1)
        P, pb, _ = ops.MatMul.apply(parts[1], Wm_ext, False, False).split([2 * H + 4, 1, 3], dim=1)   # Wu2 @ [Wm | 0 | bm]
        Wc = P + torch.cat([parts[0], ops.zeros_view(parts[0].shape[0], P.shape[1] - parts[0].shape[1],
                                                     device=P.device)], dim=1)      # (F.pad: a fill and a copy)
        bc = pb.view(-1) + bu
        if self.if_agg_story_in_update:
            return Wc, bc, torch.cat([parts[2] - parts[3], parts[3]], dim=1)
        return Wc, bc, None

    def forward_composed(self, voxel_feature, plan, pdm4, composed, grads=None, posb=None):
        """v + block(v) with the collapsed weights (valid when plan.vox_all_has).  ``grads`` as in Attention.forward;
        ``posb`` = ops.block_pos_term(pdm4, Wc, bc) when the caller evaluated it once for all calls of a forward."""
        Wc, bc, Wsb = composed
        H = voxel_feature.shape[1]
        if (self.if_agg_story_in_update and not torch.is_grad_enabled()
                and "story_building_mean" in plan.ops and H % 4 == 0 and H <= 256):
            # graph-free pass: the gather-mean, the story mean and both of its projections in one launch, then
            # term[s] = P[s] + mean over the stories of s's graph of Q  (P, Q = the two halves)
            if plan.Nv <= FUSE_GATHER_STORY_MAX_ROWS:
                gm, pq = ops.gather_story_raw(plan.vox_mean, plan.story_seg, voxel_feature, Wsb)
            else:       # large batches fill the chip on their own (10^6 voxels: 417 + 96 us as two launches, 632 us as one)
                gm = ops.spmm_raw(plan.vox_mean, voxel_feature)
                pq = ops.story_project_raw(plan.story_seg, voxel_feature, Wsb)
            term = ops.spmm_add_raw(plan.story_building_mean, pq[:, H:], pq[:, :H])
            if posb is not None:        # the product starts from the position + bias term and runs over K = 2H
                return ops.gemm_raw([voxel_feature, gm], Wc[:, :2 * H], False, True, act=ACT_LEAKY, slope=0.01,
                                    pre_add=term, pre_rows=plan.story_gather.csr.col, residual=voxel_feature,
                                    acc_init=posb)
            return linear([voxel_feature, gm, pdm4], Wc, bc, ACT_LEAKY, 0.01, pre_add=term,
                          pre_gather=plan.story_gather, residual=voxel_feature)
        if (self.if_agg_story_in_update and FUSE_GBLOCK and torch.is_grad_enabled() and "story_building_mean" in plan.ops
                and H % 4 == 0 and H <= 256 and plan.Nv <= FUSE_GATHER_STORY_MAX_ROWS):
            # a graph is being built (generator update): the whole block as one autograd node
            return ops.GBlock.apply(voxel_feature, Wc, bc, Wsb, pdm4, plan, 0.01,
                                    *(grads if grads is not None else (None, True)), PDM4_PAD == 1.0, posb)
        gm = SpMM.apply(voxel_feature, plan.vox_mean)
        if not self.if_agg_story_in_update:
            return linear([voxel_feature, gm, pdm4], Wc, bc, residual=voxel_feature)
        story = SpMM.apply(voxel_feature, plan.story_mean)                      # [S, H]
        building = SpMM.apply(SpMM.apply(story, plan.graph_of_story_mean), plan.story_graph_gather)   # [S, H]
        # (Wu3 - Wu4) story + Wu4 building, evaluated per STORY; the product's epilogue gathers it to the voxels
        term = linear([story, building], Wsb)
        return linear([voxel_feature, gm, pdm4], Wc, bc, ACT_LEAKY, 0.01, pre_add=term, pre_gather=plan.story_gather,
                      residual=voxel_feature)

    def forward(self, voxel_feature, plan, pdm4):
        H = voxel_feature.shape[1]
        gm = SpMM.apply(voxel_feature, plan.vox_mean)
        aggr = run_mlp(self.messageMLP, [voxel_feature, gm, pdm4], row_scale=plan.vox_has, pad_to=2 * H + 4)
        if not self.if_agg_story_in_update:
            return run_mlp(self.updateMLP, [voxel_feature, aggr])
        story = SpMM.apply(voxel_feature, plan.story_mean)
        building = SpMM.apply(story, plan.graph_of_story_mean)
        story_v = SpMM.apply(story, plan.story_gather)
        building_v = SpMM.apply(building, plan.graph_gather)
        return run_mlp(self.updateMLP, [voxel_feature, aggr, story_v, building_v - story_v])


class VoxelGNN_G(nn.Module):
    def __init__(self, input_dim, hidden_dim, noise_dim, layer_num, max_len: int = 20):
        super().__init__()
        self.layer_num = layer_num
        self.noise_dim = noise_dim
        self.register_buffer("pe", position_encoder(hidden_dim, max_len))
        self.enc = MLP([input_dim - 3 + noise_dim, hidden_dim])
        self.block = VoxelBlock(hidden_dim, if_agg_story_in_update=True)
        self.attention = Attention(hidden_dim)

    def derived_weights(self):
        """What every forward derives from the parameters alone: the composed block weights (Wc, bc, Wsb) and
        the attention's stacked weight (Wcat, bcat).  A caller that knows the parameters have not changed
        (train.Trainer between two generator updates) may keep them and pass them back as ``derived``."""
        Wc, bc, Wsb = self.block.compose()
        Wcat, bcat = self.attention.stacked_weights()
        return [Wc, bc, Wsb, Wcat, bcat]

    def _one_node_ok(self, x, v, plan) -> bool:
        """ops.VoxelStackG applies: the conditions of ops.GBlock / ops.AttentionNode with the position term taken once,
        and plan pieces the one-call backward can run (bg_gstack_backward_f32)."""
        H = x.shape[1]
        o = plan.ops
        if not (FUSE_GSTACK and torch.is_grad_enabled() and x.is_cuda and FUSE_GBLOCK and FUSE_ATTENTION_NODE
                and SHARE_WEIGHT_GRADS and PDM4_PAD == 1.0 and self.noise_dim == 0 and len(self.enc) == 1
                and self.block.if_agg_story_in_update and plan.vox_all_has and self.layer_num >= 1
                and H % 64 == 0 and H <= 256 and 0 < plan.Nv <= FUSE_GATHER_STORY_MAX_ROWS and plan.Np > 0
                and 0 < plan.cross_v.max_len <= 64 and "story_building_mean" in o and "prog_pair" in o
                and o["max_level"] < self.pe.shape[0] and v.shape[1] == 4 and v.dtype == torch.float32):
            return False
        st, cp = plan.story_seg, plan.cross_p
        if st.nnz != plan.Nv or st.count is None or plan.cross_v.col is None:
            return False
        return cp.row_per_workgroup or not cp.long_rows          # (two-phase reductions stay on the op-by-op path)

    def forward(self, x, v, z, plan, noise: Optional[Sequence] = None, derived=None):
        n_calls = 1 + sum(1 for i in range(self.layer_num) if i % 2 == 0 and i != 0)
        if noise is None:
            # all Exp(1) draws of this forward (mask [Nv, 2] + attention [Ec] per call) in ONE launch
            per = 2 * plan.Nv + (plan.Ec + 3) // 4 * 4           # every piece starts 16-byte aligned
            flat = torch.empty(n_calls * per, dtype=torch.float32, device=v.device).exponential_()
            noise = [(flat[c * per:c * per + 2 * plan.Nv].view(plan.Nv, 2),
                      flat[c * per + 2 * plan.Nv:c * per + 2 * plan.Nv + plan.Ec]) for c in range(n_calls)]
        pos, v = extract_pos(v)
        pdm4 = _pos_terms(plan, pos)
        if self._one_node_ok(x, v, plan):
            # a graph is being built (generator update): everything below the program GNN is ONE autograd node whose
            # forward and backward are one host call each
            att = self.attention
            Wc, bc, Wsb = tuple(derived[:3]) if derived is not None else self.block.compose()
            Wcat, bcat = (derived[3], derived[4]) if (derived is not None and derived[3] is not None) else att.stacked_weights()
            if Wcat is not None:
                _generator_plan_struct(plan)
                ms, mh, as_, ah = ops.VoxelStackG.apply(x, self.enc[0].weight, self.enc[0].bias, att.NN_p.weight, att.NN_p.bias,
                                                        Wcat, bcat, att.theta, Wc, bc, Wsb, self.pe, v, pdm4, noise, plan,
                                                        self.layer_num)
                return {"hard": mh, "soft": ms}, {"hard": ah.view(-1, 1), "soft": as_.view(-1, 1)}, None
        draws = iter(noise)
        enc = self.enc[0]                       # pe[level] rides in the encoder product's epilogue (gathered addend)
        v = linear(torch.cat((v, z), dim=-1) if self.noise_dim != 0 else v, enc.weight, enc.bias,
                   pre_add=self.pe, pre_gather=_level_op(plan, self.pe.shape[0]))
        use = derived is not None and derived[3] is not None
        shared = self.attention.shared_terms(x, (derived[3], derived[4]) if use else None)
        # the block's and the attention's calls chain through v: their nodes sum the shared weights' gradients
        # in their own launches and the first call's node hands the sum on (ops.SharedGrads)
        sg = ops.SharedGrads() if SHARE_WEIGHT_GRADS and torch.is_grad_enabled() else None
        _, _, v = self.attention(x, v, plan, next(draws), shared, (sg, True) if sg is not None else None)
        out_mask = out_att = None
        composed = None
        if plan.vox_all_has:
            composed = tuple(derived[:3]) if derived is not None else self.block.compose()
        posb = None
        if (composed is not None and self.block.if_agg_story_in_update and plan.Nv <= FUSE_GATHER_STORY_MAX_ROWS
                and "story_building_mean" in plan.ops and v.shape[1] % 4 == 0 and v.shape[1] <= 256):
            # the position columns and the bias enter every one of the 12 calls alike (same module): once
            posb = ops.block_pos_term(pdm4, composed[0], composed[1])
        for i in range(self.layer_num):
            if composed is not None:
                v = self.block.forward_composed(v, plan, pdm4, composed, (sg, i == 0) if sg is not None else None, posb)
            else:
                v = v + self.block(v, plan, pdm4)
            if i % 2 == 0 and i != 0:
                out_mask, out_att, v = self.attention(x, v, plan, next(draws), shared,
                                                      (sg, False) if sg is not None else None)
        return out_mask, out_att, v


class VoxelGNN_D(nn.Module):
    def __init__(self, input_feature_dim, input_label_dim, hidden_dim, layer_num, max_len: int = 20):
        super().__init__()
        self.layer_num = layer_num
        self.register_buffer("pe", position_encoder(hidden_dim, max_len))
        self.feature_enc = MLP([input_feature_dim - 3, hidden_dim])
        self.label_enc = MLP([input_label_dim, hidden_dim])
        self.block = VoxelBlock(2 * hidden_dim, if_agg_story_in_update=False)
        self._share_coef = False      # see DiscriminatorVoxel.shared_coefficients
        self._coef_cache = None
        self.collapse_ok = True       # cleared by DiscriminatorVoxel.guard_collapse when the polynomial form is ill-conditioned

    # ---- collapsed evaluation ---------------------------------------------------------------------
    # Neither messageMLP nor updateMLP of the discriminator's block has an activation (reference
    # models.py:149-150 with if_agg_story_in_update=False), the encoders are single Linears
    # (models.py:211-212) and the 12 layers share one block (models.py:213,227-228).  On graphs where
    # every voxel has an in-edge the whole GNN is therefore the affine map
    #     v_L = sum_k A^k U T_k ,   U = [feature | onehot(level) | mean dpos | 1 | label]  (N x ~40)
    # with A the mean-aggregation operator and T_k (~40 x 2H) polynomial coefficients obtained from the
    # parameters by the recurrence T^{m+1}_k = T^m_k (I + Wa^T) + T^m_{k-1} Wb^T + [k=0] E.
    # One pass costs L thin gather-means (8 label columns; the label-free columns are shared by every
    # pass of an iteration) and ONE [N, (L+1)*40] x [.., 2H] GEMM instead of L GEMMs of K = 4H+4.
    # Basis (ops.D_BASIS): the same polynomial is carried as  v_L = sum_k T_k(A) U C_k  with the Chebyshev
    # polynomials T_k (T_{k+1} = 2 x T_k - T_{k-1}; the chain kernels run this three-term form) and the table's
    # recurrence C^{m+1} = C^m (I + Wa^T) + (x * C^m) Wb^T + [k=0] E, where  x * C  is bg_cheb_shift_f32.  A's
    # spectrum lies in [-1, 1], where |T_k| <= 1: the entries of C stay of the size of the polynomial's values, while
    # the monomial coefficients T_k grow (and cancel) as the parameters train -- measured in DESIGN.md section 6.
    def _static_powers(self, plan, feat, pdm4, rider=None):
        """[N, (L+1)*ds]: A^k [feature | onehot(level) | mean dpos | 1 | 0..] for k = 0..L (label-free
        columns; no gradient).  Kept on the plan until Trainer / the caller starts the next iteration."""
        key = ("d_static_powers", self.layer_num, self.pe.shape[0], ops.D_BASIS)
        Z = plan.ops.get(key)
        if Z is None:
            ML = self.pe.shape[0]
            _level_op(plan, ML)                                       # raises when a building has more stories than pe rows
            width = feat.shape[1] + ML + 4 + 1
            ds = (width + 3) // 4 * 4
            # column group 0 in one launch; the chain of operator powers writes groups 1..L
            Z = torch.empty(feat.shape[0], (self.layer_num + 1) * ds, device=feat.device)
            ops.d_static_cols_raw(feat.detach(), plan.level, ML, pdm4, Z, ds)
            if rider is not None:      # (Zk, width): another matrix whose powers are due -- both in one launch
                ops.power_chain_pair_raw(plan.vox_mean, Z, ds, rider[0], rider[1], self.layer_num, plan.vox_blocks)
                rider[2] = True
            else:
                ops.power_chain_raw(plan.vox_mean, Z, self.layer_num, ds, plan.vox_blocks)
            plan.ops[key] = Z
        return Z

    def coefficients(self, n_feat, ds, dl):
        """The collapsed weight, shared between calls inside DiscriminatorVoxel.shared_coefficients()."""
        key = (ds, dl, torch.is_grad_enabled(), ops.D_BASIS)
        if self._share_coef and self._coef_cache is not None and self._coef_cache[0] == key:
            return self._coef_cache[1]
        W = self._table_from_parameters(n_feat, ds, dl)
        if W is None:
            W = self._collapsed_weight(self.block.compose(), n_feat, ds, dl)
        if self._share_coef:
            self._coef_cache = (key, W)
        return W

    def _table_from_parameters(self, n_feat, ds, dl):
        """compose() + PolyCoefD as one node (ops.PolyCoefDW), or None where that form does not apply."""
        blk, fe, le = self.block, self.feature_enc[0], self.label_enc[0]
        Wm, Wu = blk.messageMLP[0].weight, blk.updateMLP[0].weight
        H2, L = Wm.shape[0], self.layer_num
        H = H2 // 2
        if not (FUSE_COMPOSE_D and FUSE_COEF_INPUTS and ops.FUSE_COEF_RECURRENCE and Wm.is_cuda and L >= 1
                and not blk.if_agg_story_in_update and Wm.shape[1] == 2 * H2 + 3 and tuple(Wu.shape) == (H2, 2 * H2)
                and fe.weight.shape == (H, n_feat) and le.weight.shape[0] == H and H2 % 4 == 0):
            return None
        XL = ops.PolyCoefDW.apply(Wm, blk.messageMLP[0].bias, Wu, blk.updateMLP[0].bias, fe.weight, fe.bias, le.weight,
                                  le.bias, self.pe, L, ds, dl)
        return SpMM.apply(XL, self._coef_row_order(L, ds, dl, XL.device))

    def _table_manual(self, n_feat, ds, dl):
        """_table_from_parameters without an autograd graph: (W, finish) where finish(gW) returns the gradients of
        (messageMLP weight, bias, updateMLP weight, bias, feature_enc weight, bias, label_enc weight, bias) -- the same
        two nodes' forward and backward called directly (DiscriminatorVoxel.critic_objective(manual=True))."""
        blk, fe, le = self.block, self.feature_enc[0], self.label_enc[0]
        Wm, Wu = blk.messageMLP[0].weight, blk.updateMLP[0].weight
        H2, L = Wm.shape[0], self.layer_num
        H = H2 // 2
        if not (FUSE_COMPOSE_D and FUSE_COEF_INPUTS and ops.FUSE_COEF_RECURRENCE and Wm.is_cuda and L >= 1
                and not blk.if_agg_story_in_update and Wm.shape[1] == 2 * H2 + 3 and tuple(Wu.shape) == (H2, 2 * H2)
                and fe.weight.shape == (H, n_feat) and le.weight.shape[0] == H and H2 % 4 == 0):
            return None
        params = (Wm, blk.messageMLP[0].bias, Wu, blk.updateMLP[0].bias, fe.weight, fe.bias, le.weight, le.bias)
        ctx = ops.ManualCtx(13)
        XL = ops.PolyCoefDW.forward(ctx, *[t.detach() for t in params], self.pe.detach(), L, ds, dl)
        order = self._coef_row_order(L, ds, dl, XL.device)
        W = ops.spmm_raw(order, XL)

        def finish(gW):
            return params, ops.PolyCoefDW.backward(ctx, ops.spmm_raw(order.T, gW))[:8]
        return W, finish

    def _coef_row_order(self, L, ds, dl, device) -> LinOp:
        key = (L, ds, dl, str(device))
        cache = self.__dict__.setdefault("_row_order", {})
        op = cache.get(key)
        if op is None:
            from .plan import Csr
            du, n = ds + dl, (L + 1) * (ds + dl)
            k = torch.arange(L + 1).view(-1, 1) * du
            src = torch.cat([(k + torch.arange(ds)).reshape(-1), (k + ds + torch.arange(dl)).reshape(-1)])
            inv = torch.empty(n, dtype=torch.int64)
            inv[src] = torch.arange(n)
            rowptr = iota_i32(n + 1, device)
            op = LinOp.pair(Csr(rowptr, src.to(torch.int32).to(device), None, n, n, n),
                            Csr(rowptr, inv.to(torch.int32).to(device), None, n, n, n))
            cache[key] = op
        return op

    def _collapsed_weight(self, composed, n_feat, ds, dl):
        """W [(L+1)*ds + (L+1)*dl, 2H] such that v_L = [Zs | Zl] W (rows ordered (k, input column))."""
        Wc, bc = composed[0], composed[1]
        H2 = Wc.shape[0]
        H, ML, L, du = H2 // 2, self.pe.shape[0], self.layer_num, ds + dl
        fe, le = self.feature_enc[0], self.label_enc[0]
        if (FUSE_COEF_INPUTS and fe.weight.shape[0] == H and le.weight.shape[0] == H and Wc.shape[1] == 2 * H2 + 4
                and fe.weight.shape[1] == n_feat):
            # X0 / E written into the recurrence's buffers by one launch, their gradients read out by one
            XL = ops.PolyCoefD.apply(Wc, bc, fe.weight, fe.bias, le.weight, le.bias, self.pe, L, ds, dl)
            # rows (k, column) -> all static columns first, then all label columns: a row permutation as one
            # gather launch (its backward: the inverse gather) instead of split / two reshape copies / cat
            return SpMM.apply(XL, self._coef_row_order(L, ds, dl, XL.device))
        z = lambda r, c: ops.zeros_view(r, c, device=Wc.device)     # cached zeros: no fill launches
        pad_s = ds - (n_feat + ML + 4 + 1)
        n_lab = le.weight.shape[1]
        # T^0_0 [du, 2H]: how the input columns enter v_0 = [feature_enc(f) + pe[level] | label_enc(l)]
        X0 = torch.cat([
            torch.cat([fe.weight.t(), z(n_feat, H)], dim=1),
            torch.cat([self.pe, z(ML, H)], dim=1),
            z(4, H2),
            torch.cat([fe.bias, le.bias]).view(1, -1),
            z(pad_s, H2),
            torch.cat([z(n_lab, H), le.weight.t()], dim=1),
            z(dl - n_lab, H2)], dim=0)
        # E [du, 2H]: what every layer adds: mean-dpos columns through Wc's position block, the bias
        E = torch.cat([z(n_feat + ML, H2), Wc[:, 2 * H2:2 * H2 + 4].t(), bc.view(1, -1), z(pad_s + dl, H2)], dim=0)
        Ts, Tl = ops.PolyCoef.apply(Wc, X0, E, L).view(L + 1, du, H2).split([ds, dl], dim=1)
        return torch.cat([Ts.reshape(-1, H2), Tl.reshape(-1, H2)], dim=0)

    def collapse_amplification(self, feat, l, plan, pdm4, sample: int = 2048):
        """How much the collapsed evaluation v_L = [Zs | Zl] W amplifies fp32 rounding on THIS batch with the
        CURRENT parameters: max over output columns of  max_i (|Z| |W|)_io / max_i |Z W|_io  on a strided sample
        of the voxel rows -- the classical bound on a product's rounding error is eps * (|Z| |W|), so the
        collapsed form's error relative to a column's largest activation is ~ eps * this number (device scalar; no
        host sync).  It grows with ||I + Wa|| ^ L when terms of the polynomial cancel: see DiscriminatorVoxel.guard_collapse."""
        L = self.layer_num
        with torch.no_grad():
            Zs = self._static_powers(plan, feat, pdm4)
            ds = Zs.shape[1] // (L + 1)
            dl = (l.shape[1] + 3) // 4 * 4
            W = self.coefficients(feat.shape[1], ds, dl)
            Zl = torch.zeros(l.shape[0], L + 1, dl, device=l.device)
            Zl[:, 0, :l.shape[1]] = l.detach()
            ops.power_chain_raw(plan.vox_mean, Zl, L, dl, plan.vox_blocks)
            step = max(1, feat.shape[0] // sample)
            Z = torch.cat([Zs[::step], Zl.view(Zl.shape[0], -1)[::step]], dim=1).contiguous()
            v = linear([Z], W, w_t=True)
            bound = linear([Z.abs()], W.abs().contiguous(), w_t=True)
            return (bound.amax(0) / v.abs().amax(0).clamp_min(1e-30)).amax()

    def forward_collapsed(self, feat, l, plan, pdm4):
        L = self.layer_num
        Zs = self._static_powers(plan, feat, pdm4)
        ds = Zs.shape[1] // (L + 1)
        dl = (l.shape[1] + 3) // 4 * 4
        W = self.coefficients(feat.shape[1], ds, dl)
        zl = ops.PowerChain.apply(F.pad(l, (0, dl - l.shape[1])), plan.vox_mean, L, plan.vox_blocks)
        return linear([Zs, zl.view(zl.shape[0], -1)], W, w_t=True)

    def _label_powers(self, feat, labels, plan, pdm4):
        """(Zs [N, (L+1) ds], Zk [N, L+1, P dl], W, ds, dl): operator powers of the label-free columns and of the P
        label sets (data), and the coefficient table."""
        L, P, N = self.layer_num, len(labels), feat.shape[0]
        n_lab = labels[0].shape[1]
        dl = (n_lab + 3) // 4 * 4
        Zk = torch.empty(N, L + 1, P * dl, device=feat.device)       # group 0 packed here, the rest by the chain
        ops.pack_columns_raw([l.detach() for l in labels], dl, Zk.view(N, -1))
        rider = [Zk, P * dl, False]      # when the label-free powers are still to be made, the label powers ride along
        Zs = self._static_powers(plan, feat, pdm4, rider)
        ds = Zs.shape[1] // (L + 1)
        W = self.coefficients(feat.shape[1], ds, dl)
        if not rider[2]:
            ops.power_chain_raw(plan.vox_mean, Zk, L, P * dl, plan.vox_blocks)
        return Zs, Zk, W, ds, dl

    def collapsed_multi(self, feat, labels, plan, pdm4):
        """v_L of several label sets that share the graph (the passes of one critic update), labels
        taken as data (no gradient reaches them through this call).  The label-free part S = Zs Ws is
        evaluated once; each pass adds its own thin product.  Returns ([v_L per pass], Wl, dl) where
        Wl [(L+1)*dl, 2H] maps label powers to the output (rows ordered (k, label column))."""
        L, P, N = self.layer_num, len(labels), feat.shape[0]
        Zs, Zk, W, ds, dl = self._label_powers(feat, labels, plan, pdm4)
        Ws, Wl = W.split([(L + 1) * ds, (L + 1) * dl], dim=0)
        Zp = Zk.view(N, L + 1, P, dl).permute(0, 2, 1, 3).contiguous().view(N, P, (L + 1) * dl)
        S = linear([Zs], Ws, w_t=True)
        return [linear([Zp[:, p]], Wl, pre_add=S, w_t=True) for p in range(P)], Wl, dl

    def forward(self, v, l, plan):
        pos, v = extract_pos(v)
        pdm4 = _pos_terms(plan, pos)
        if COLLAPSE_D and self.collapse_ok and plan.vox_all_has and self.layer_num > 0:
            return self.forward_collapsed(v, l, plan, pdm4)
        v = run_mlp(self.feature_enc, v)
        v = v + ops.spmm_raw(_level_op(plan, self.pe.shape[0]), self.pe)
        v = torch.cat([v, run_mlp(self.label_enc, l)], dim=-1)
        composed = self.block.compose() if plan.vox_all_has else None
        if composed is not None and COMPOSE_STACK and self.layer_num > 0:
            return ops.AffineStack.apply(v, composed[0], composed[1], pdm4, plan.vox_mean, self.layer_num)
        for _ in range(self.layer_num):
            if composed is not None:
                v = self.block.forward_composed(v, plan, pdm4, composed)
            else:
                v = v + self.block(v, plan, pdm4)
        return v


def _generator_plan_struct(plan):
    """include/bgan.h bgGeneratorPlan of a plan (kept on the plan: it only holds addresses of the plan's own tensors)."""
    from . import _lib
    ops_ = plan.ops
    ps = ops_.get("_generator_plan_struct")
    if ps is None:
        pp, vx, st, sb, cx = (ops_["prog_pair"], plan.vox_mean, plan.story_seg, ops_["story_building_mean"], plan.cross_v)
        P = _lib.ptr
        ps = _lib.GeneratorPlan(plan.Np, plan.Nv, plan.S, plan.Ec, cx.max_len, 1,
                                P(pp.csr.rowptr), P(pp.csr.col), P(pp.row_scale), P(plan.prog_has),
                                P(vx.csr.rowptr), P(vx.csr.col), P(vx.row_scale),
                                P(st.rowptr), P(st.col), P(st.count), P(plan.story_gather.csr.col),
                                P(sb.csr.rowptr), P(sb.csr.col), P(sb.row_scale), P(plan.level),
                                P(cx.rowptr), P(cx.col))
        ops_["_generator_plan_struct"] = ps
    return ps


class Generator(nn.Module):
    def __init__(self, program_input_dim, voxel_input_dim, hidden_dim, noise_dim, program_layer, voxel_layer,
                 device=None, max_len: int = 20):
        super().__init__()
        self.device = device
        self.hidden_dim = hidden_dim
        self.noise_dim = noise_dim
        self.programGNN = ProgramGNN(program_input_dim, hidden_dim, noise_dim, program_layer)
        self.voxelGNN = VoxelGNN_G(voxel_input_dim, hidden_dim, 0, voxel_layer, max_len)

    def _forward_one_call(self, graph, plan, program_z, noise, derived):
        """The graph-free forward through bg_generator_forward_f32: the same launches as the op-by-op path below,
        issued by one host call.  None when a condition of that entry point does not hold (the caller then runs the
        op-by-op path)."""
        from . import _lib
        import ctypes as C
        gnn, att, pg = self.voxelGNN, self.voxelGNN.attention, self.programGNN
        H, L = self.hidden_dim, gnn.layer_num
        ops_ = plan.ops
        if not (plan.vox_all_has and 0 < plan.cross_v.max_len <= 64 and H % 64 == 0 and H <= 256 and L >= 1
                and 0 < plan.Nv <= FUSE_GATHER_STORY_MAX_ROWS and plan.Np > 0 and "story_building_mean" in ops_
                and "prog_pair" in ops_ and gnn.noise_dim == 0 and gnn.block.if_agg_story_in_update
                and ops_["flags"]["cross_by_voxel"][1] == 0 and graph.program_class_feature.shape[1] <= 8
                and plan.ops["max_level"] < gnn.pe.shape[0] and len(pg.enc) == 1 and len(gnn.enc) == 1):
            return None
        if derived is None:
            derived = gnn.derived_weights()
        Wc, bc, Wsb, Wcat, bcat = derived
        if Wcat is None or Wc.stride(1) != 1 or not (bc.is_contiguous() and Wsb.is_contiguous() and Wcat.is_contiguous()
                                                     and bcat.is_contiguous()):
            return None
        dev = graph.voxel_feature.device
        n_cls = graph.program_class_feature.shape[1]
        pos, v4 = extract_pos(graph.voxel_feature)
        pdm4 = _pos_terms(plan, pos)
        program_in = torch.cat([graph.program_class_feature, graph.story_level_feature.view(-1, 1), program_z], dim=-1)
        if program_in.dtype != torch.float32 or v4.dtype != torch.float32:
            return None
        n_calls = 1 + sum(1 for i in range(L) if i % 2 == 0 and i != 0)
        Nv, Ec = plan.Nv, plan.Ec
        if noise is None:
            per = 2 * Nv + (Ec + 3) // 4 * 4
            flat = torch.empty(n_calls * per, dtype=torch.float32, device=dev).exponential_()
            base = flat.data_ptr()
            em_ptrs = [base + 4 * c * per for c in range(n_calls)]
            ea_ptrs = [base + 4 * (c * per + 2 * Nv) for c in range(n_calls)]
            keep = flat
        else:
            noise = list(noise)
            if len(noise) < n_calls:
                return None
            keep = [(ops._f32c(a).contiguous(), ops._f32c(b.reshape(-1)).contiguous()) for a, b in noise[:n_calls]]
            if any(tuple(a.shape) != (Nv, 2) or b.numel() != Ec for a, b in keep):
                raise _lib.BganError("generator: noise must hold one ([Nv, 2], [Ec]) pair per Attention call")
            em_ptrs, ea_ptrs = [a.data_ptr() for a, _ in keep], [b.data_ptr() for _, b in keep]
        lib = _lib.load()
        ps = _generator_plan_struct(plan)
        penc, venc, blk = pg.enc[0], gnn.enc[0], pg.block
        tensors = [penc.weight, penc.bias, blk.messageMLP[0].weight, blk.messageMLP[0].bias, blk.updateMLP[0].weight,
                   blk.updateMLP[0].bias, venc.weight, venc.bias, gnn.pe, Wc, bc, Wsb, Wcat, bcat, att.NN_p.weight,
                   att.NN_p.bias, att.theta]
        if any(t.dtype != torch.float32 or t.device != dev for t in tensors) or any(
                not t.is_contiguous() for i, t in enumerate(tensors) if i != 9):
            return None
        d = [t.data_ptr() for t in tensors]
        ws = _lib.GeneratorWeights(H, pg.enc[0].weight.shape[1] - n_cls - 1, pg.layer_num, L, n_cls, gnn.pe.shape[0],
                                   d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], Wc.stride(0),
                                   d[10], d[11], d[12], d[13], d[14], d[15], d[16])
        if program_in.shape[1] != ws.n_class + 1 + ws.noise_dim:
            return None
        nws = int(lib.bg_generator_forward_workspace(plan.Np, Nv, plan.S, Ec, H))
        work = torch.empty(nws, dtype=torch.float32, device=dev)
        out = torch.empty(Nv, n_cls, dtype=torch.float32, device=dev)
        soft_out = torch.empty(Nv, n_cls, dtype=torch.float32, device=dev)
        mask_hard = torch.empty(Nv, 1, dtype=torch.float32, device=dev)
        mask_soft = torch.empty(Nv, 1, dtype=torch.float32, device=dev)
        att_hard = torch.empty(Ec, 1, dtype=torch.float32, device=dev)
        att_soft = torch.empty(Ec, 1, dtype=torch.float32, device=dev)
        idx = torch.empty(Nv, dtype=torch.int64, device=dev)
        PA = _lib.ptr_array_type(n_calls)
        cls = ops._f32c(graph.program_class_feature).contiguous()
        rc = lib.bg_generator_forward_f32(C.byref(ws), C.byref(ps), program_in.data_ptr(), v4.data_ptr(), pdm4.data_ptr(),
                                          cls.data_ptr(), PA(*em_ptrs), PA(*ea_ptrs), n_calls, work.data_ptr(), nws,
                                          out.data_ptr(), soft_out.data_ptr(), mask_hard.data_ptr(), mask_soft.data_ptr(),
                                          att_hard.data_ptr(), att_soft.data_ptr(), idx.data_ptr(), _lib.stream())
        del keep
        if rc == -2:                      # BG_ERR_SHAPE: not a shape for the one-call path
            return None
        _lib.check(rc, "bg_generator_forward_f32")
        return out, soft_out, {"hard": mask_hard, "soft": mask_soft}, {"hard": att_hard, "soft": att_soft}, idx

    def forward(self, graph, program_z, voxel_z, noise: Optional[Sequence] = None, derived=None):
        plan = build_plan(graph)
        if FUSE_GENERATOR_FORWARD and not torch.is_grad_enabled() and graph.voxel_feature.is_cuda:
            res = self._forward_one_call(graph, plan, program_z, noise, derived)
            if res is not None:
                return res
        x = self.programGNN(graph.program_class_feature, graph.story_level_feature, graph.program_target_ratio,
                            program_z, plan)
        mask, att, _ = self.voxelGNN(x, graph.voxel_feature, voxel_z, plan, noise, derived)
        out, soft_out, idx = Attention.construct_output(graph.program_class_feature, graph.voxel_feature.shape[0],
                                                        att, mask, plan, graph.cross_edge_program_index_select)
        return out, soft_out, mask, att, idx


def pool_story_and_graph(v, plan):
    """global_max_pool of the voxel features per story and per graph (reference models.py:262-263).
    With no empty story the graph maximum is taken over the [S, H] story maxima (exactly the same
    values; one pass over the voxels instead of two)."""
    story, _ = ops.SegMax.apply(v, plan.story_seg)
    if plan.stories_all_nonempty:
        building, _ = ops.SegMax.apply(story, plan.story_graph_seg)
    else:
        building, _ = ops.SegMax.apply(v, plan.graph_seg)
    return story, building


class DiscriminatorVoxel(nn.Module):
    def __init__(self, voxel_input_feature_dim, voxel_input_label_dim, hidden_dim, voxel_layer, act="",
                 max_len: int = 20):
        super().__init__()
        self.voxelGNN_D = VoxelGNN_D(voxel_input_feature_dim, voxel_input_label_dim, hidden_dim, voxel_layer, max_len)
        self.building_dec = MLP([2 * hidden_dim, hidden_dim, 1], act)
        self.story_dec = MLP([2 * hidden_dim, hidden_dim, 1], act)
        # no activation anywhere (the WGAN-GP configuration): outputs depend on the forward
        # activations only through arg-max choices, so a gradient penalty reaches the parameters
        # through the double-backward graph alone (see ops.input_grads_only)
        self.piecewise_linear = not any(k in act for k in ("relu", "tanh", "sigmoid"))

    @contextlib.contextmanager
    def shared_coefficients(self):
        """The caller promises not to change the parameters inside this context (the passes of one
        discriminator update: fake, real, gradient-penalty sample): the collapsed GNN's coefficient
        table -- a function of the parameters alone -- is then built once and shared by the passes,
        and its backward runs once on the summed gradient."""
        gnn = self.voxelGNN_D
        gnn._share_coef, gnn._coef_cache = True, None
        try:
            yield self
        finally:
            gnn._share_coef, gnn._coef_cache = False, None

    def critic_terms(self, graph, label_real, label_fake, label_hat=None, stacked: bool = False):
        """Every discriminator evaluation of one critic update (reference train.py:88-96) in one
        pass over the collapsed GNN: D(real), D(fake) and, when ``label_hat`` is given, the gradients
        of D(hat)'s two outputs with respect to ``label_hat`` (reference losses.py:83-84) -- obtained
        in closed form, sum_k (A^T)^k G Wl_k^T with G the decoders' gradient scattered to the arg-max
        voxels, as a differentiable function of the parameters -- instead of two autograd passes
        over a third forward graph.  Labels are treated as data.

        Returns None when the collapsed form does not apply to this graph (the caller then runs the
        passes one by one), else (real_b, real_s, fake_b, fake_s, grad_b, grad_s); with ``stacked`` the
        same numbers un-split: (val_b [2B,1] real rows first, val_s [2S,1], grads [N, 2, dl] or None) where
        grads[:, 0] / grads[:, 1] are grad_b / grad_s zero-padded to dl columns."""
        plan = build_plan(graph)
        gnn = self.voxelGNN_D
        if not (COLLAPSE_D and gnn.collapse_ok and plan.vox_all_has and gnn.layer_num > 0):
            return None
        pos, feat = extract_pos(graph.voxel_feature)
        pdm4 = _pos_terms(plan, pos)
        labels = [label_real, label_fake] + ([label_hat] if label_hat is not None else [])
        vs, Wl, dl = gnn.collapsed_multi(feat, labels, plan, pdm4)
        B, S, N = plan.B, plan.S, plan.Nv
        # real and fake share one run of each decoder
        pools = [pool_story_and_graph(v, plan) for v in vs[:2]]
        val_s_all = run_mlp(self.story_dec, torch.cat([pools[0][0], pools[1][0]], dim=0))
        val_b_all = run_mlp(self.building_dec, torch.cat([pools[0][1], pools[1][1]], dim=0))
        if stacked:
            out = None
        else:
            val_s, val_b = val_s_all.split(S, dim=0), val_b_all.split(B, dim=0)
            out = [val_b[0], val_s[0], val_b[1], val_s[1]]
        if label_hat is None:
            return (val_b_all, val_s_all, None) if stacked else (*out, None, None)
        n_lab, L = label_hat.shape[1], gnn.layer_num
        with torch.no_grad():
            hat_s, arg_s = ops.segmax_raw(plan.story_seg, vs[2])
            if plan.stories_all_nonempty:
                hat_b, arg_sb = ops.segmax_raw(plan.story_graph_seg, hat_s)       # arg-max story of each graph ..
                # .. and that story's arg-max voxel: the int32 indices travel through the gather as raw bits
                arg_b = ops.gather_arg_raw(arg_s.view(torch.float32), arg_sb).view(torch.int32)
            else:
                hat_b, arg_b = ops.segmax_raw(plan.graph_seg, vs[2])
        gpools = []
        for dec, pooled in ((self.building_dec, hat_b), (self.story_dec, hat_s)):
            pooled = pooled.requires_grad_(True)
            with ops.input_grads_only():
                val = run_mlp(dec, pooled)
            gpools.append(torch.autograd.grad(val, pooled, torch.ones_like(val), create_graph=True)[0])
        G2 = ops.ScatterArgPair.apply(gpools[0], arg_b, gpools[1], arg_s, N,       # [2N, 2H]: building rows first
                                      plan.graph_seg.seg_of, plan.story_seg.seg_of)
        X = ops.MatMul.apply(G2, Wl, False, True)                                  # [2N, (L+1)*dl]
        # sum_k (A^T)^k Y_k over both outputs' columns side by side: one launch on block-diagonal plans
        Ys = X.view(2, N, L + 1, dl).permute(1, 2, 0, 3).reshape(N, L + 1, 2 * dl)
        acc = ops.HornerChain.apply(Ys, plan.vox_mean.T, plan.vox_blocks)
        if stacked:
            return val_b_all, val_s_all, acc.view(N, 2, dl)      # the pad columns are exactly zero
        grad_b, _, grad_s, _ = acc.split([n_lab, dl - n_lab, n_lab, dl - n_lab], dim=1)
        return (*out, grad_b, grad_s)

    def critic_objective(self, graph, label_real, label_fake, label_hat, wb, ws, coef: float, manual: bool = False):
        """(objective, bare penalty sum) of one WGAN-GP critic update -- <D(real), D(fake), wb / ws> + coef * sum over
        voxels and both outputs of (||d D(hat) / d label|| - 1)^2 -- with everything below the coefficient table issued
        by ONE host call each way (ops.CriticHead) instead of critic_terms' ~90 launches from the interpreter.  Same
        kernels, same formulas (tested against critic_terms + CriticLoss).  Returns None where it does not apply
        (activation in a decoder, empty stories, blocks too large for the chain kernel, ...): the caller then takes
        critic_terms."""
        plan = build_plan(graph)
        gnn = self.voxelGNN_D
        if not (FUSE_CRITIC_HEAD and COLLAPSE_D and gnn.collapse_ok and plan.vox_all_has
                and gnn.layer_num > 0 and self.piecewise_linear and label_hat is not None):
            return None
        blocks, st, sg = plan.vox_blocks, plan.story_seg, plan.story_graph_seg
        if (blocks is None or not plan.stories_all_nonempty or not st.row_per_workgroup or not sg.row_per_workgroup
                or len(self.building_dec) != 2 or len(self.story_dec) != 2 or blocks.max_deg > 8):
            return None
        pos, feat = extract_pos(graph.voxel_feature)
        pdm4 = _pos_terms(plan, pos)
        bd, sd = self.building_dec, self.story_dec
        dec = (bd[0].weight, bd[0].bias, bd[1].weight, bd[1].bias, sd[0].weight, sd[0].bias, sd[1].weight, sd[1].bias)
        if manual:
            # ``manual``: the update's three nodes (coefficient table, row order, head) called forward and backward
            # right here, their gradients assigned to ``.grad`` -- no autograd graph, no engine pass (its thread hop,
            # one AccumulateGrad per parameter): ~0.1 ms of host time per critic update, same kernels in the same order
            with torch.no_grad():
                L, n_lab = gnn.layer_num, label_real.shape[1]
                dl = (n_lab + 3) // 4 * 4
                Zk = torch.empty(feat.shape[0], L + 1, 3 * dl, device=feat.device)
                ops.pack_columns_raw([label_real.detach(), label_fake.detach(), label_hat.detach()], dl, Zk.view(feat.shape[0], -1))
                rider = [Zk, 3 * dl, False]
                Zs = gnn._static_powers(plan, feat, pdm4, rider)
                ds = Zs.shape[1] // (L + 1)
                made = gnn._table_manual(feat.shape[1], ds, dl)
                if made is None:
                    return None
                W, finish = made
                if not rider[2]:
                    ops.power_chain_raw(plan.vox_mean, Zk, L, 3 * dl, plan.vox_blocks)
                if not ops._chain_fits(plan.vox_mean.T, blocks, 2 * dl) or W.shape[1] % 4 != 0:
                    return None
                ctx = ops.ManualCtx(18)
                total, pen = ops.CriticHead.forward(ctx, W.contiguous(), *[t.detach() for t in dec], Zs, Zk, plan, wb, ws,
                                                    coef, L, ds, dl)
                grads = ops.CriticHead.backward(ctx, ops.ones_view(1, device=W.device), None)
                for prm, g in zip(dec, grads[1:9]):
                    prm.grad = g
                for prm, g in zip(*finish(grads[0])):
                    prm.grad = g
                if gnn._share_coef:       # the generator objective of this iteration re-uses the table (it is taken
                    gnn._coef_cache = ((ds, dl, False, ops.D_BASIS), W)       # without a graph there)
            return total, pen
        Zs, Zk, W, ds, dl = gnn._label_powers(feat, [label_real, label_fake, label_hat], plan, pdm4)
        if not ops._chain_fits(plan.vox_mean.T, blocks, 2 * dl) or W.shape[1] % 4 != 0:
            return None
        return ops.CriticHead.apply(W.contiguous(), *dec, Zs, Zk, plan, wb, ws, coef, gnn.layer_num, ds, dl)

    def generator_objective(self, graph, label, wb, ws):
        """<D_building(label), wb> + <D_story(label), ws> for the generator update (the discriminator frozen: only
        ``label`` is differentiated), everything below the coefficient table by ONE host call each way
        (ops.GLossHead).  None where it does not apply; the caller then calls forward() and forms the sums."""
        plan = build_plan(graph)
        gnn = self.voxelGNN_D
        if not (FUSE_CRITIC_HEAD and COLLAPSE_D and gnn.collapse_ok and plan.vox_all_has
                and gnn.layer_num > 0 and self.piecewise_linear and not any(p.requires_grad for p in self.parameters())):
            return None
        blocks, st, sg = plan.vox_blocks, plan.story_seg, plan.story_graph_seg
        if (blocks is None or not plan.stories_all_nonempty or not st.row_per_workgroup or not sg.row_per_workgroup
                or len(self.building_dec) != 2 or len(self.story_dec) != 2 or blocks.max_deg > 8):
            return None
        pos, feat = extract_pos(graph.voxel_feature)
        pdm4 = _pos_terms(plan, pos)
        L = gnn.layer_num
        dl = (label.shape[1] + 3) // 4 * 4
        with torch.no_grad():
            Zs = gnn._static_powers(plan, feat, pdm4)
            ds = Zs.shape[1] // (L + 1)
            W = gnn.coefficients(feat.shape[1], ds, dl)
        if not ops._chain_fits(plan.vox_mean, blocks, dl) or W.shape[1] % 4 != 0:
            return None
        bd, sd = self.building_dec, self.story_dec
        return ops.GLossHead.apply(label, W.contiguous(), bd[0].weight, bd[0].bias, bd[1].weight, bd[1].bias, sd[0].weight,
                                   sd[0].bias, sd[1].weight, sd[1].bias, Zs, plan, wb, ws, L, ds, dl)

    # fp32 error of the collapsed form relative to a column's largest activation ~ 2^-24 * amplification (measured:
    # tests/test_gpu_models.py::test_collapsed_discriminator_guard); beyond this bound the polynomial form would lose
    # more than north_star's 1e-4 and the layer-by-layer stack (ops.AffineStack: the reference's own evaluation order)
    # takes over until a later check finds the parameters back inside
    COLLAPSE_AMPLIFICATION_LIMIT = 1e-4 / 2.0 ** -24 / 8.0

    def guard_collapse(self, graph, label=None, reduce_max=None) -> float:
        """Runtime guard of the collapsed (polynomial) evaluation: measure its rounding amplification on this batch
        with the current parameters (one small product on <= 2,048 sampled voxels; ONE host read, so call it every
        few hundred optimizer steps -- train.Trainer does, `collapse_check_every`) and switch the discriminator
        between the collapsed form and the layer-by-layer stack accordingly.  Returns the amplification.
        ``reduce_max`` (a one-element-tensor -> tensor function, e.g. an all-reduce MAX over the data-parallel group)
        makes the decision the same on every rank: each rank measures its own batch, all of them act on the largest
        value (NaN counts as +inf), so no rank evaluates the discriminator in another order than its peers."""
        plan = build_plan(graph)
        gnn = self.voxelGNN_D
        applies = bool(COLLAPSE_D and plan.vox_all_has and gnn.layer_num > 0)
        if not applies and reduce_max is None:
            return 0.0
        if applies:
            pos, feat = extract_pos(graph.voxel_feature)
            label = graph.voxel_label if label is None else label
            amp_t = gnn.collapse_amplification(feat, label, plan, _pos_terms(plan, pos)).detach().reshape(1).float()
            amp_t = torch.where(torch.isnan(amp_t), torch.full_like(amp_t, float("inf")), amp_t)
        else:                                     # this rank's batch takes the stack anyway: it still joins the collective
            amp_t = torch.zeros(1, dtype=torch.float32, device=graph.voxel_feature.device)
        if reduce_max is not None:
            amp_t = reduce_max(amp_t)
        amp = float(amp_t)
        gnn.collapse_ok = amp <= self.COLLAPSE_AMPLIFICATION_LIMIT                       # +inf (a NaN somewhere) -> stack
        return amp

    def prepare(self, graph, n_label: int = NUMBER_OF_CLASS):
        """Inside shared_coefficients(): build what the collapsed passes share and that depends on
        the graph and the parameters only (operator powers of the label-free columns, coefficient
        table), e.g. on a side stream while the generator runs."""
        plan = build_plan(graph)
        gnn = self.voxelGNN_D
        if not (COLLAPSE_D and gnn.collapse_ok and plan.vox_all_has and gnn.layer_num > 0 and gnn._share_coef):
            return
        pos, feat = extract_pos(graph.voxel_feature)
        Zs = gnn._static_powers(plan, feat, _pos_terms(plan, pos))
        gnn.coefficients(feat.shape[1], Zs.shape[1] // (gnn.layer_num + 1), (n_label + 3) // 4 * 4)

    def forward(self, graph, out_label):
        plan = build_plan(graph)
        v = self.voxelGNN_D(graph.voxel_feature, out_label, plan)
        story, building = pool_story_and_graph(v, plan)
        return run_mlp(self.building_dec, building), run_mlp(self.story_dec, story)
