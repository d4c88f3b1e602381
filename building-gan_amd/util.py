"""Host mirror of /root/reference/util.py on the HIP kernels (same names and argument meaning)."""
from __future__ import annotations

import torch

from . import ops


def softmax_to_hard(y_soft: torch.Tensor, dim: int) -> torch.Tensor:
    """reference util.py:6-10: one-hot of the arg-max, straight-through gradient of ``y_soft``."""
    winner = y_soft.max(dim, keepdim=True)[1]
    y_hard = torch.zeros_like(y_soft).scatter_(dim, winner, 1.0)
    return y_hard - y_soft.detach() + y_soft


def gumbel_softmax(src: torch.Tensor, plan, tau: float = -1, exp_noise: torch.Tensor = None):
    """reference util.py:13-28 over the cross-edge segments of ``plan`` (segments = voxels).

    The reference takes the COO ``index`` vector; here the segment structure comes from the
    batch's plan (``plan.cross_v``), which was built from that same vector.
    """
    if exp_noise is None:
        exp_noise = torch.empty_like(src).exponential_()
    soft, hard, _ = ops.SegGumbelSoftmax.apply(src, exp_noise, plan, float(tau))
    return soft, hard
