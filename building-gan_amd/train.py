"""One G+D training iteration with the reference's schedule (/root/reference/train.py:73-114),
flat fp32 parameter / gradient buffers, a HIP Adam step, and data parallelism by graph batch:
one process per GPU, ONE sum all-reduce (RCCL over xGMI) of the flat gradient buffer per network
per optimizer step -- D every iteration (331,010 floats), G every n_critic_g-th (228,418).

The reference has no multi-GPU path (util_graph.py:149-158 is marked "Not used"); the loss terms
are means over graphs, stories and voxels, so with ``world_size > 1`` each rank divides its local
sums by the GLOBAL counts (all-reduced once per batch) and gradients are summed: the result equals
a single-device step over the union batch.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist

from . import ops
from .graph import detach_batch
from .losses import VolumetricDesignLoss_G
from .models import DiscriminatorVoxel, Generator
from .util import softmax_to_hard


class FlatAdam:
    """Parameters and gradients of a module as views into two flat buffers + HIP Adam."""

    def __init__(self, module: torch.nn.Module, lr=1e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = [p for p in module.parameters()]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.view(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = 0

    def zero_grad(self):
        self.grad.zero_()
        off = 0
        for p in self.params:                      # autograd may have replaced .grad; re-attach the views
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view_as(p)
            off += k

    def all_reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def step(self):
        self.t += 1
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t)


def global_counts(batch, group=None):
    """(graphs, stories, voxels) over all ranks -- one tiny all-reduce per batch."""
    B = int(batch.FAR.shape[0])
    Nv = int(batch.voxel_feature.shape[0])
    plan = batch.__dict__.get("_bg_plan")
    S = plan.S if plan is not None else int(batch.voxel_floor_cluster.max().item()) + 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.tensor([B, S, Nv], dtype=torch.float64, device=batch.voxel_feature.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        B, S, Nv = (int(x) for x in t.tolist())
    return B, S, Nv


class Trainer:
    def __init__(self, device, hidden_dim=128, noise_dim=32, program_layer=4, voxel_layer=12,
                 g_lr=1e-4, d_lr=1e-4, b1=0.5, b2=0.999, gp_lambda=10.0, n_critic_d=1, n_critic_g=5,
                 gan_loss="WGANGP", max_len=20, group=None):
        if gan_loss != "WGANGP":
            raise NotImplementedError("the data-parallel step implements the default WGANGP objective")
        self.device = device
        self.noise_dim = noise_dim
        self.generator = Generator(7, 7, hidden_dim, noise_dim, program_layer, voxel_layer, device, max_len).to(device)
        self.discriminator = DiscriminatorVoxel(7, 6, hidden_dim, voxel_layer, act="", max_len=max_len).to(device)
        self.opt_g = FlatAdam(self.generator, g_lr, (b1, b2))
        self.opt_d = FlatAdam(self.discriminator, d_lr, (b1, b2))
        self.gp_lambda, self.n_critic_d, self.n_critic_g = gp_lambda, n_critic_d, n_critic_g
        self.g_loss_fn = VolumetricDesignLoss_G(gan_loss=gan_loss)
        self.group = group

    def broadcast_parameters(self, src: int = 0):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.broadcast(self.opt_g.flat, src, group=self.group)
            dist.broadcast(self.opt_d.flat, src, group=self.group)

    @staticmethod
    def _mean(x, n_global):
        return x.sum() / n_global

    def _gradient_penalty(self, batch, label, out, u, Nv_g):
        """reference losses.py:73-89 with the voxel mean taken over the global voxel count."""
        if u is None:
            u = torch.rand(label.shape[0], 1, device=out.device)
        mixed = (label * u + out * (1 - u)).detach().requires_grad_(True)
        nonzero = (mixed.max(dim=-1)[0] != 0).type(torch.float32).view(-1, 1)
        sample = softmax_to_hard(mixed, -1) * nonzero
        val_b, val_s = self.discriminator(batch, sample)
        grad_b = torch.autograd.grad(val_b, sample, torch.ones_like(val_b), retain_graph=True, create_graph=True)[0]
        grad_s = torch.autograd.grad(val_s, sample, torch.ones_like(val_s), retain_graph=True, create_graph=True)[0]
        gp_b = ((grad_b.norm(2, 1) - 1) ** 2).sum() / Nv_g
        gp_s = ((grad_s.norm(2, 1) - 1) ** 2).sum() / Nv_g
        return gp_b + gp_s

    def train_iteration(self, batch, i: int, program_z=None, voxel_z=None, noise: Optional[Sequence] = None,
                        u: Optional[torch.Tensor] = None):
        """Iteration ``i`` of an epoch (reference train.py:74-114).  Returns a dict of detached losses."""
        G, D = self.generator, self.discriminator
        dev = self.device
        stats = {}
        for p in self.opt_d.params:
            p.requires_grad = True
        self.opt_d.zero_grad()
        if program_z is None:
            program_z = torch.rand(batch.program_class_feature.shape[0], self.noise_dim, device=dev)
        if voxel_z is None:
            voxel_z = torch.rand(batch.voxel_feature.shape[0], self.noise_dim, device=dev)
        out, soft_out, mask, att, max_out_program_index = G(batch, program_z, voxel_z, noise)
        Bg, Sg, Nvg = global_counts(batch, self.group)

        if i % self.n_critic_d == 0:
            fake_b, fake_s = D(detach_batch(batch), out.detach())
            real_b, real_s = D(batch, batch.voxel_label)
            d_loss = (-self._mean(real_b, Bg) - self._mean(real_s, Sg) + self._mean(fake_b, Bg) + self._mean(fake_s, Sg))
            if self.gp_lambda != 0:
                gp = self._gradient_penalty(detach_batch(batch), batch.voxel_label.detach(), soft_out.detach(), u, Nvg)
                d_loss = d_loss + self.gp_lambda * gp
                stats["gp"] = gp.detach()
            d_loss.backward()
            self.opt_d.all_reduce(self.group)
            self.opt_d.step()
            stats["d_loss"] = d_loss.detach()

        for p in self.opt_d.params:
            p.requires_grad = False
        self.opt_g.zero_grad()
        if i % self.n_critic_g == 0:
            val_b, val_s = D(batch, out)
            # adversarial term with global means; auxiliary terms have zero weight by default (train_args.py:40-42)
            self.g_loss_fn(( val_b, val_s), batch, att["hard"], mask["hard"], 6)   # runs get_program_ratio as the reference does
            g_loss = -self._mean(val_b, Bg) - self._mean(val_s, Sg)
            g_loss.backward()
            self.opt_g.all_reduce(self.group)
            self.opt_g.step()
            stats["g_loss"] = g_loss.detach()
        return stats
