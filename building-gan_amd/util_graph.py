"""Host mirror of the hot-path helpers of /root/reference/util_graph.py on the HIP kernels."""
from __future__ import annotations

import torch

from . import ops
from .graph import detach_batch  # noqa: F401  (reference util_graph.py:8-18)
from .ops import SpMM
from .plan import build_plan


def get_program_ratio(graph, att, mask, area_index_in_voxel_feature):
    """reference util_graph.py:21-41: masked voxel area -> cross edge -> program node ->
    (graph, class) slot -> graph.  Returns (normalised class weight [6B,1], program weight
    [Np,1], FAR [B,1])."""
    plan = build_plan(graph)
    vf = graph.voxel_feature
    if (vf.is_cuda and not (torch.is_grad_enabled() and (att.requires_grad or mask.requires_grad or vf.requires_grad))
            and vf.dtype == torch.float32 and vf.stride(1) == 1):
        got = ops.program_ratio_raw(plan, att, mask, vf, area_index_in_voxel_feature)      # values only: two launches
        if got is not None:
            return got
    masked_voxel_weight = mask * graph.voxel_feature[:, area_index_in_voxel_feature].view(-1, 1)
    painted = att * SpMM.apply(masked_voxel_weight, plan.edge_voxel_gather)
    program_weight = SpMM.apply(painted, plan.edge_program_gather.T)
    program_class_weight = SpMM.apply(program_weight, plan.slot_sum)
    far = SpMM.apply(program_class_weight, plan.slot_graph_sum)
    batch_sum = SpMM.apply(far, plan.slot_graph_gather)
    return program_class_weight / (batch_sum + 1e-16), program_weight, far


def find_max_out_program_index(logit, graph):
    """reference util_graph.py:44-48 (first arg-max cross edge of every voxel -> its program node)."""
    plan = build_plan(graph)
    arg = ops.seg_argmax(logit, plan)
    return graph.cross_edge_program_index_select[arg.long()]
