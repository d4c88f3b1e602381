"""Oracle restatement of the pointer / ratio helpers of the hot path (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/util.py:6-28 (straight-through hard one-hot; segment Gumbel-softmax with
its tau = -1 default) and /root/reference/util_graph.py:21-48 (program ratio pipeline, arg-max
program index).  Randomness is an explicit input: wherever the reference draws Exp(1) noise the
caller passes the pre-drawn tensor (None -> draw it as the reference does), so the HIP path and
the oracle can be fed identical noise.
"""
from __future__ import annotations

from typing import Optional

import torch

from .scatter import maybe_num_nodes, scatter_max, scatter_sum


def straight_through_one_hot(soft: torch.Tensor, dim: int) -> torch.Tensor:
    """reference util.py:6-10 (softmax_to_hard): value one_hot(argmax) - soft + soft, grad of soft."""
    winner = soft.max(dim, keepdim=True)[1]
    hard = torch.zeros_like(soft).scatter_(dim, winner, 1.0)
    return hard - soft.detach() + soft


softmax_to_hard = straight_through_one_hot


def dense_gumbel_softmax(logits: torch.Tensor, exp_noise: Optional[torch.Tensor] = None,
                         tau: float = 1.0) -> torch.Tensor:
    """torch.nn.functional.gumbel_softmax(logits) at its defaults (tau=1, hard=False, dim=-1):
    softmax((logits - log(Exp(1))) / tau); used at reference models.py:109."""
    e = torch.empty_like(logits).exponential_() if exp_noise is None else exp_noise
    return ((logits + (-e.log())) / tau).softmax(-1)


def segment_gumbel_softmax(src: torch.Tensor, index: torch.Tensor, tau: float = -1,
                           num_nodes: Optional[int] = None, exp_noise: Optional[torch.Tensor] = None):
    """reference util.py:13-28 (gumbel_softmax): softmax over the elements sharing ``index``.

    y = (src - log(Exp(1))) / tau with tau = -1 by default (so the winner is the arg-MIN of the
    perturbed logits); soft = exp(y - segmax(y)) / (segsum + 1e-16); hard = one-hot of the
    first arg-max of soft per segment, with the straight-through value (hard - soft) + soft.
    """
    n = maybe_num_nodes(index, num_nodes)
    e = torch.empty_like(src).exponential_() if exp_noise is None else exp_noise
    y = (src + (-e.log())) / tau
    shifted = (y - scatter_max(y, index, n)[0][index]).exp()
    soft = shifted / (scatter_sum(shifted, index, n)[index] + 1e-16)
    winner = scatter_max(soft, index, n)[1]
    hard = torch.zeros_like(soft).scatter_(0, winner, 1.0)
    return soft, hard - soft.detach() + soft


gumbel_softmax = segment_gumbel_softmax


def find_max_out_program_index(logit, cross_voxel, cross_program, num_voxels):
    """reference util_graph.py:44-48: program node of the arg-max cross edge of every voxel."""
    return cross_program[scatter_max(logit, cross_voxel, num_voxels)[1]]


def get_program_ratio(graph, att, mask, area_col: int):
    """reference util_graph.py:21-41: masked voxel area -> cross edge -> program node ->
    (graph, class) slot -> graph; returns (normalised class ratio, program weight, FAR)."""
    cev, cep = graph.cross_edge_voxel_index_select, graph.cross_edge_program_index_select
    slot_graph = graph.program_target_ratio_batch
    n_graph = graph.FAR.shape[0]
    area = mask * graph.voxel_feature[:, area_col].view(-1, 1)
    on_edge = att * area.index_select(0, cev)
    per_program = scatter_sum(on_edge, cep, graph.program_class_feature.shape[0])
    per_slot = scatter_sum(per_program, graph.program_class_cluster, graph.program_target_ratio.shape[0])
    per_graph = scatter_sum(per_slot, slot_graph, n_graph)
    ratio = per_slot / (per_graph[slot_graph] + 1e-16)
    return ratio, per_program, per_graph
