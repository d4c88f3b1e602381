"""GAN objectives with the reference's interface (/root/reference/Model/losses.py:8-89).

Pure torch glue over the HIP-backed networks; ``compute_gradient_penalty`` differentiates the
discriminator twice, which the op family of ops.py supports.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .util import softmax_to_hard
from .util_graph import get_program_ratio


class VolumetricDesignLoss_D(nn.Module):
    def __init__(self, gan_loss, gp_lambda=10):
        super().__init__()
        self.gan_loss = gan_loss
        self.gp_lambda = gp_lambda

    def forward(self, real_validity_voxel, fake_validity_voxel, gp=None):
        rb, rs = real_validity_voxel
        fb, fs = fake_validity_voxel
        if self.gan_loss == "WGANGP":
            loss = -rb.mean() - rs.mean() + fb.mean() + fs.mean()
            return loss + (self.gp_lambda * gp if gp is not None else 0)
        if self.gan_loss == "NSGAN":
            bce = F.binary_cross_entropy
            return (bce(rb, torch.ones_like(rb)) + bce(rs, torch.ones_like(rs))
                    + bce(fb, torch.zeros_like(fb)) + bce(fs, torch.zeros_like(fs)))
        if self.gan_loss == "LSGAN":
            mse = F.mse_loss
            return 0.5 * (mse(rb, torch.ones_like(rb)) + mse(rs, torch.ones_like(rs))
                          + mse(fb, torch.zeros_like(fb)) + mse(fs, torch.zeros_like(fs)))
        if self.gan_loss == "hinge":
            return F.relu(1.0 - rb).mean() + F.relu(1.0 - rs).mean() + F.relu(fb + 1.0).mean() + F.relu(fs + 1.0).mean()
        raise TypeError("self.gan_loss is not valid")


class VolumetricDesignLoss_G(nn.Module):
    def __init__(self, lp_weight=0, tr_weight=0, far_weight=0, embedding_dim=128, sample_size=0, similarity_fun="",
                 gan_loss="WGANGP", lp_loss="", hinge_margin=0):
        super().__init__()
        self.gan_loss = gan_loss
        self.tr_weight, self.lp_weight, self.far_weight = tr_weight, lp_weight, far_weight

    def forward(self, fake_validity_voxel, graph, att, mask, area_index_in_voxel_feature):
        fb, fs = fake_validity_voxel
        if self.gan_loss in ("WGANGP", "hinge"):
            adversarial_loss = -fb.mean() - fs.mean()
        elif self.gan_loss == "NSGAN":
            adversarial_loss = F.binary_cross_entropy(fb, torch.ones_like(fb)) + F.binary_cross_entropy(fs, torch.ones_like(fs))
        elif self.gan_loss == "LSGAN":
            adversarial_loss = F.mse_loss(fb, torch.ones_like(fb)) + F.mse_loss(fs, torch.ones_like(fs))
        else:
            raise TypeError("self.gan_loss is not valid")
        ratio, _, far = get_program_ratio(graph, att, mask, area_index_in_voxel_feature)
        zero = torch.zeros(1, device=fb.device)
        tr = zero if self.tr_weight == 0 else self.tr_weight * F.smooth_l1_loss(ratio.flatten(), graph.program_target_ratio)
        fl = zero if self.far_weight == 0 else self.far_weight * F.smooth_l1_loss(far.view(far.shape[0]), graph.FAR)
        return adversarial_loss + tr + fl, adversarial_loss, tr, fl


def compute_gradient_penalty(Dv, batch, label, out, u: Optional[torch.Tensor] = None):
    """reference losses.py:73-89; ``u`` (U(0,1) [Nv,1]) may be supplied for reproducibility."""
    if u is None:
        u = torch.rand(label.shape[0], 1).to(out.device)      # CPU generator, as the reference
    mixed = (label * u + out * (1 - u)).detach().requires_grad_(True)
    nonzero = (mixed.max(dim=-1)[0] != 0).type(torch.float32).view(-1, 1)
    sample = softmax_to_hard(mixed, -1) * nonzero
    if getattr(Dv, "piecewise_linear", False):
        with ops.input_grads_only():
            val_b, val_s = Dv(batch, sample)
    else:
        val_b, val_s = Dv(batch, sample)
    grad_b = torch.autograd.grad(val_b, sample, torch.ones_like(val_b), retain_graph=True, create_graph=True)[0]
    grad_s = torch.autograd.grad(val_s, sample, torch.ones_like(val_s), retain_graph=True, create_graph=True)[0]
    gp_b = ((grad_b.norm(2, 1) - 1) ** 2).mean()
    gp_s = ((grad_s.norm(2, 1) - 1) ** 2).mean()
    return gp_b + gp_s, gp_b, gp_s
