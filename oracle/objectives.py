"""Oracle restatement of the GAN objectives and of one training iteration (TEST INFRASTRUCTURE).

Follows /root/reference/Model/losses.py:8-89 (D loss, G loss, WGAN-GP gradient penalty with a
double backward through D) and the iteration schedule of /root/reference/train.py:73-114.
The uniform interpolation weights ``u`` and all generator noise are explicit inputs.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from .pointer import get_program_ratio, straight_through_one_hot


def d_loss_fn(real_validity, fake_validity, gp=None, gan_loss: str = "WGANGP", gp_lambda: float = 10.0):
    """reference losses.py:14-35."""
    rb, rs = real_validity
    fb, fs = fake_validity
    if gan_loss == "WGANGP":
        loss = -rb.mean() - rs.mean() + fb.mean() + fs.mean()
        return loss + (gp_lambda * gp if gp is not None else 0)
    ones_b, ones_s = torch.ones_like(rb), torch.ones_like(rs)
    zeros_b, zeros_s = torch.zeros_like(fb), torch.zeros_like(fs)
    if gan_loss == "NSGAN":
        return (F.binary_cross_entropy(rb, ones_b) + F.binary_cross_entropy(rs, ones_s)
                + F.binary_cross_entropy(fb, zeros_b) + F.binary_cross_entropy(fs, zeros_s))
    if gan_loss == "LSGAN":
        return 0.5 * (F.mse_loss(rb, ones_b) + F.mse_loss(rs, ones_s) + F.mse_loss(fb, zeros_b) + F.mse_loss(fs, zeros_s))
    if gan_loss == "hinge":
        return (F.relu(1.0 - rb).mean() + F.relu(1.0 - rs).mean() + F.relu(fb + 1.0).mean() + F.relu(fs + 1.0).mean())
    raise TypeError("self.gan_loss is not valid")


def g_loss_fn(fake_validity, graph, att, mask, area_col: int = 6, gan_loss: str = "WGANGP",
              tr_weight: float = 0.0, far_weight: float = 0.0):
    """reference losses.py:46-70; returns (total, adversarial, target-ratio, FAR) losses."""
    fb, fs = fake_validity
    if gan_loss in ("WGANGP", "hinge"):
        adv = -fb.mean() - fs.mean()
    elif gan_loss == "NSGAN":
        adv = F.binary_cross_entropy(fb, torch.ones_like(fb)) + F.binary_cross_entropy(fs, torch.ones_like(fs))
    elif gan_loss == "LSGAN":
        adv = F.mse_loss(fb, torch.ones_like(fb)) + F.mse_loss(fs, torch.ones_like(fs))
    else:
        raise TypeError(gan_loss)
    ratio, _, far = get_program_ratio(graph, att, mask, area_col)
    zero = torch.zeros(1, device=fb.device)
    tr = zero if tr_weight == 0 else tr_weight * F.smooth_l1_loss(ratio.flatten(), graph.program_target_ratio)
    fl = zero if far_weight == 0 else far_weight * F.smooth_l1_loss(far.view(far.shape[0]), graph.FAR)
    return adv + tr + fl, adv, tr, fl


def compute_gradient_penalty(Dv, batch, label, out, u: Optional[torch.Tensor] = None):
    """reference losses.py:73-89.  ``u`` = the U(0,1) [Nv,1] interpolation weights."""
    if u is None:
        u = torch.rand(label.shape[0], 1, device=out.device)
    mixed = (label * u + out * (1 - u)).detach().requires_grad_(True)
    nonzero = (mixed.max(dim=-1)[0] != 0).type(torch.float32).view(-1, 1)
    sample = straight_through_one_hot(mixed, -1) * nonzero
    val_b, val_s = Dv(batch, sample)
    grad_b = torch.autograd.grad(val_b, sample, torch.ones_like(val_b), retain_graph=True, create_graph=True)[0]
    grad_s = torch.autograd.grad(val_s, sample, torch.ones_like(val_s), retain_graph=True, create_graph=True)[0]
    gp_b = ((grad_b.norm(2, 1) - 1) ** 2).mean()
    gp_s = ((grad_s.norm(2, 1) - 1) ** 2).mean()
    return gp_b + gp_s, gp_b, gp_s


def train_iteration(generator, discriminator, opt_g, opt_d, batch, detach_batch, i: int, program_z, voxel_z,
                    noise: Optional[Sequence], u: Optional[torch.Tensor], n_critic_d: int = 1, n_critic_g: int = 5,
                    gp_lambda: float = 10.0, gan_loss: str = "WGANGP", tr_weight: float = 0.0, far_weight: float = 0.0):
    """One iteration of reference train.py:73-114 (index ``i`` inside the epoch)."""
    stats = {}
    for p in discriminator.parameters():
        p.requires_grad = True
    opt_d.zero_grad()
    out, soft_out, mask, att, idx = generator(batch, program_z, voxel_z, noise)
    if i % n_critic_d == 0:
        fake = discriminator(detach_batch(batch), out.detach())
        real = discriminator(batch, batch.voxel_label)
        gp = None
        if gan_loss == "WGANGP" and gp_lambda != 0:
            gp, _, _ = compute_gradient_penalty(discriminator, detach_batch(batch), batch.voxel_label.data,
                                                soft_out.data, u)
        d_loss = d_loss_fn(real, fake, gp, gan_loss, gp_lambda)
        d_loss.backward()
        opt_d.step()
        stats["d_loss"] = d_loss.detach()
        stats["gp"] = gp.detach() if gp is not None else None
    for p in discriminator.parameters():
        p.requires_grad = False
    opt_g.zero_grad()
    if i % n_critic_g == 0:
        validity = discriminator(batch, out)
        g_loss, _, _, _ = g_loss_fn(validity, batch, att["hard"], mask["hard"], 6, gan_loss, tr_weight, far_weight)
        g_loss.backward()
        opt_g.step()
        stats["g_loss"] = g_loss.detach()
    return stats
