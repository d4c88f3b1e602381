"""One G+D training iteration with the reference's schedule (/root/reference/train.py:73-114),
flat fp32 parameter / gradient buffers, a HIP Adam step, and data parallelism by graph batch:
one process per GPU, ONE sum all-reduce (RCCL over xGMI) of the flat gradient buffer per network
per optimizer step -- D every iteration (331,010 floats), G every n_critic_g-th (228,418).

The reference has no multi-GPU path (util_graph.py:149-158 is marked "Not used"); the loss terms
are means over graphs, stories and voxels, so with ``world_size > 1`` each rank divides its local
sums by the GLOBAL counts (all-reduced once per batch) and gradients are summed: the result equals
a single-device step over the union batch.

For fixed-shape batches the iteration can be captured into hipGraphs (``Trainer.capture``): the
compute between two collectives is one graph, the all-reduce stays an eager RCCL call between
graph launches, so the same structure runs on 1..8 GPUs.
"""
from __future__ import annotations

import contextlib
from typing import Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops
from .graph import detach_batch
from .util_graph import get_program_ratio
from .models import DiscriminatorVoxel, Generator, position_terms
from .util import softmax_to_hard


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


# With one rank every collective is the identity and is skipped.  FORCE_COLLECTIVES = True issues them anyway when a
# process group exists (bench.py --force-dist, tests/test_rccl_one_rank.py): the RCCL call path -- all-reduce of the
# flat gradient buffers, broadcast of the parameters, the counts' collective on the side stream, the cuts of the
# captured iteration at the collectives -- then runs on a single GPU exactly as it does with eight.
FORCE_COLLECTIVES = False


def _collective(group=None) -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return FORCE_COLLECTIVES or dist.get_world_size(group) > 1


class FlatAdam:
    """Parameters and gradients of a module as views into two flat buffers + HIP Adam."""

    def __init__(self, module: torch.nn.Module, lr=1e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = [p for p in module.parameters()]
        # every parameter starts on a 256-byte boundary: the GEMM's 16-byte vector loads need aligned
        # weights, and the pad (zero gradient, zero moments) stays zero under Adam
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.view(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = torch.zeros(1, dtype=torch.int32, device=dev) if dev.type == "cuda" else None
        self.t_host = 0

    def _grad_views(self):
        """The flat gradient buffer seen through each parameter's shape (made once: three tensor ops per parameter)."""
        v = self.__dict__.get("_views")
        if v is None:
            v = self._views = [self.grad[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self.offsets)]
        return v

    def zero_grad(self):
        self.grad.zero_()
        for p, view in zip(self.params, self._grad_views()):   # autograd may have replaced .grad; re-attach the views
            if p.grad is not view:
                p.grad = view

    def detach_grads(self):
        """Before a backward pass whose gradients ``collect`` will gather: with ``.grad`` unset autograd hands
        every parameter its gradient tensor as is, instead of one add launch per parameter into the views of a
        freshly zeroed flat buffer."""
        for p in self.params:
            p.grad = None

    def collect(self):
        """All parameter gradients into the flat buffer by ONE concatenation (pads and parameters without a
        gradient read a cached zero block); ``.grad`` is a view of the flat buffer again afterwards."""
        dev = self.grad.device
        pads = self.__dict__.get("_pads")
        if pads is None:          # zero blocks between / after the parameters, and one per parameter without a gradient
            pads, end = [], 0
            for p, off in zip(self.params, self.offsets):
                pads.append((ops.zeros_view(off - end, device=dev) if off > end else None,
                             ops.zeros_view(p.numel(), device=dev)))
                end = off + p.numel()
            self._pads = pads
            self._tail = ops.zeros_view(self.grad.numel() - end, device=dev) if self.grad.numel() > end else None
        pieces = []
        for p, (gap, zero) in zip(self.params, pads):
            if gap is not None:
                pieces.append(gap)
            g = p.grad
            pieces.append(zero if g is None else (g.detach() if g.requires_grad else g).reshape(-1))
        if self._tail is not None:
            pieces.append(self._tail)
        with torch.no_grad():
            torch.cat(pieces, out=self.grad)
        for p, view in zip(self.params, self._grad_views()):
            p.grad = view

    def all_reduce(self, group=None, exchange: str = "all_reduce"):
        """Sum the flat gradient buffer over the ranks.  ``exchange``: "all_reduce" = the backend's all-reduce (RCCL's
        default algorithm choice); "gather_sum" = the one-shot form SURVEY section 5 names for these latency-bound 1 MB
        messages: ONE all-gather (every rank sends its buffer to every peer over its own xGMI link at once), then a
        local sum over the gathered copies in rank order -- one communication step instead of a ring's 2 (N - 1), and a
        summation order that is the same on every rank and in every run."""
        if _collective(group):
            if self.grad.is_cuda and dist.get_backend(group) == "gloo":
                torch.cuda.current_stream().synchronize()      # gloo reads the buffer from the host side
            if exchange == "gather_sum":
                w = _world(group)
                buf = self.__dict__.get("_gathered")
                if buf is None or buf.shape[0] != w:
                    buf = self._gathered = torch.empty(w, self.grad.numel(), dtype=self.grad.dtype, device=self.grad.device)
                dist.all_gather_into_tensor(buf.view(-1), self.grad, group=group)
                torch.sum(buf, dim=0, out=self.grad)           # rank order 0 .. N-1 on every rank
            elif exchange == "all_reduce":
                dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            else:
                raise ValueError("exchange must be 'all_reduce' or 'gather_sum'")

    def _need_device(self, what):
        if self.t is None:
            from ._lib import BganError
            raise BganError(f"FlatAdam.{what}: the parameters live on {self.flat.device}; the HIP library is the only "
                            "compute path of building-gan_amd (a CPU Trainer can form gradients, not step)")

    def step(self):
        self._need_device("step")
        self.t_host += 1
        self.t += 1                                # device-side count: valid under hipGraph replay
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                      0, self.t)

    def state(self):
        self._need_device("state")
        return [self.flat.clone(), self.m.clone(), self.v.clone(), self.t.clone(), self.t_host]

    def restore(self, st):
        self._need_device("restore")
        self.flat.copy_(st[0]); self.m.copy_(st[1]); self.v.copy_(st[2]); self.t.copy_(st[3])
        self.t_host = st[4]


def _local_counts(batch):
    B = int(batch.FAR.shape[0])
    Nv = int(batch.voxel_feature.shape[0])
    plan = batch.__dict__.get("_bg_plan")
    S = plan.S if plan is not None else getattr(batch, "num_stories", None)
    if S is None:
        S = int(batch.voxel_floor_cluster.max().item()) + 1
    return B, S, Nv


def global_counts(batch, group=None):
    """(graphs, stories, voxels) over all ranks -- one tiny all-reduce per batch, or what prefetch_counts() already
    fetched for this batch beside the previous iteration."""
    cached = batch.__dict__.pop("_bg_counts", None) if hasattr(batch, "__dict__") else None
    if cached is not None and cached[0] == _world(group):
        if cached[2] is not None:
            cached[2].synchronize()       # the copy to pinned memory, queued an iteration ago: finished long since
        return tuple(int(x) for x in cached[1].tolist())
    B, S, Nv = _local_counts(batch)
    if _collective(group):
        t = torch.tensor([B, S, Nv], dtype=torch.float64, device=batch.voxel_feature.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        B, S, Nv = (int(x) for x in t.tolist())
    return B, S, Nv


_PINNED_RING = []


def _pinned_counts():
    """A 3-element pinned buffer out of a small ring (page-locking memory per call would cost more than the copy;
    a buffer is read one iteration after it is filled -- prefetch_counts takes two per call, one to send the local counts
    and one to receive the sums -- and the ring holds sixteen)."""
    if not _PINNED_RING:
        _PINNED_RING.extend([torch.empty(3, dtype=torch.float64, pin_memory=True) for _ in range(16)] + [0])
    k = _PINNED_RING[-1]
    _PINNED_RING[-1] = (k + 1) % 16
    return _PINNED_RING[k]


def prefetch_counts(batch, group=None):
    """The global counts of a batch that a LATER iteration will use, fetched now on the plan's side stream: reading the
    all-reduced counts back at the start of an iteration would first drain everything the previous iteration queued
    (with one rank there is nothing to fetch).  Every rank must call this for the same batches in the same order."""
    if not _collective(group) or not hasattr(batch, "__dict__"):
        return
    from .plan import prefetch_stream
    dev = batch.voxel_feature.device
    B, S, Nv = _local_counts(batch)
    if dev.type != "cuda":
        t = torch.tensor([B, S, Nv], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        batch.__dict__["_bg_counts"] = (_world(group), t, None)
        return
    side = prefetch_stream(dev)
    with torch.cuda.stream(side):             # (the counts come from host-side shapes: nothing of the batch is read)
        # through pinned memory: torch.tensor(list, device=...) is a BLOCKING copy, and behind the plan kernels queued on
        # this stream the host waited ~0.16 ms per step for it (tools/stream_host_profile.py, FORCE_DIST=1)
        staged = _pinned_counts()
        staged[0], staged[1], staged[2] = float(B), float(S), float(Nv)
        t = torch.empty(3, dtype=torch.float64, device=dev)
        t.copy_(staged, non_blocking=True)
        if dist.get_backend(group) == "gloo":
            side.synchronize()                # gloo reads the buffer from the host side
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        # the host does NOT wait here: the sums travel to pinned memory behind the collective, an event marks the
        # copy, and global_counts() -- one iteration later -- finds it finished (as plan.prefetch_plan's flags)
        host = _pinned_counts()
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    batch.__dict__["_bg_counts"] = (_world(group), host, ev)


class Trainer:
    def __init__(self, device, hidden_dim=128, noise_dim=32, program_layer=4, voxel_layer=12,
                 g_lr=1e-4, d_lr=1e-4, b1=0.5, b2=0.999, gp_lambda=10.0, n_critic_d=1, n_critic_g=5,
                 gan_loss="WGANGP", max_len=20, group=None, generator=None, discriminator=None,
                 tr_weight=0.0, far_weight=0.0, collapse_check_every=200):
        if gan_loss not in ("WGANGP", "NSGAN", "LSGAN", "hinge"):
            raise TypeError("gan_loss is not valid")          # as reference losses.py:35
        self.gan_loss = gan_loss
        self.device = device
        self.noise_dim = noise_dim
        # networks can be injected (same call signature) -- the multi-process CPU tests do that
        self.generator = generator if generator is not None else \
            Generator(7, 7, hidden_dim, noise_dim, program_layer, voxel_layer, device, max_len).to(device)
        self.discriminator = discriminator if discriminator is not None else \
            DiscriminatorVoxel(7, 6, hidden_dim, voxel_layer, act="sigmoid" if gan_loss == "NSGAN" else "",
                               max_len=max_len).to(device)           # reference train.py:65
        self.opt_g = FlatAdam(self.generator, g_lr, (b1, b2))
        self.opt_d = FlatAdam(self.discriminator, d_lr, (b1, b2))
        # the penalty belongs to WGAN-GP only (reference train.py:92-95)
        self.gp_lambda = gp_lambda if gan_loss == "WGANGP" else 0.0
        self.n_critic_d, self.n_critic_g = n_critic_d, n_critic_g
        # auxiliary generator losses (reference losses.py:65-68, train_args.py:40-42: both default to 0)
        self.tr_weight, self.far_weight = float(tr_weight), float(far_weight)
        self.program_ratio_fn = get_program_ratio      # (graph, att, mask, area column) -> (ratio, program weight, FAR)
        # every so many eager iterations the discriminator re-measures the rounding amplification of its collapsed
        # (polynomial) evaluation under the current parameters and falls back to the layer-by-layer stack when it
        # would lose more than the fp32 tolerance (models.DiscriminatorVoxel.guard_collapse; one host read)
        self.collapse_check_every = collapse_check_every
        self.manual_critic = True            # the fused critic update without an autograd pass (critic_objective(manual=True))
        self.grad_exchange = "all_reduce"    # or "gather_sum": see FlatAdam.all_reduce
        self.group = group
        self._graphs = None
        self._captured_batch = None
        self._cw_cache = None
        # composed generator weights (functions of the parameters alone) kept between generator updates for the
        # graph-free forwards of critic-only iterations; refreshed IN PLACE (captured graphs hold the buffers)
        # after every change of the generator's parameters that goes through this class or load_state_dict
        self._g_derived = None
        if hasattr(self.generator, "register_load_state_dict_post_hook"):
            self.generator.register_load_state_dict_post_hook(lambda _m, _k: self.refresh_derived())
        self._overlap_stream = None    # set while capture() records the iteration

    def broadcast_parameters(self, src: int = 0):
        if _collective(self.group):
            dist.broadcast(self.opt_g.flat, src, group=self.group)
            dist.broadcast(self.opt_d.flat, src, group=self.group)
            self.refresh_derived()

    def _derived(self):
        """The generator's parameter-only tensors for a graph-free forward (built on first use)."""
        gnn = getattr(self.generator, "voxelGNN", None)
        if not hasattr(gnn, "derived_weights"):
            return None
        if self._g_derived is None:
            with torch.no_grad():
                self._g_derived = [None if t is None else t.clone() for t in gnn.derived_weights()]
        return self._g_derived

    def refresh_derived(self):
        """Recompute the kept tensors after the generator's parameters changed (optimizer step, broadcast,
        load_state_dict, restore).  Call it after changing the parameters in any other way."""
        if self._g_derived is None:
            return
        with torch.no_grad():
            for buf, t in zip(self._g_derived, self.generator.voxelGNN.derived_weights()):
                if buf is not None:
                    # an elementwise KERNEL, not copy_(): a contiguous device-to-device copy_ is a memcpy node under
                    # stream capture, and memset / memcpy nodes were seen to run out of order in replayed graphs on
                    # this stack (csrc/common.h, bg_zero_async); x * 1 keeps every bit pattern of finite values
                    torch.mul(t, 1.0, out=buf)

    # ------------------------------------------------------------------ objectives
    # reference losses.py:14-35 / 52-61 with every mean written as (local sum) / (global count), so that the
    # SUM all-reduce of the ranks' gradients is the gradient of the union batch.
    def _d_adversarial(self, real_b, real_s, fake_b, fake_s, Bg, Sg):
        gl = self.gan_loss
        if gl == "WGANGP":
            return -real_b.sum() / Bg - real_s.sum() / Sg + fake_b.sum() / Bg + fake_s.sum() / Sg
        if gl == "NSGAN":
            bce = lambda x, t: F.binary_cross_entropy(x, torch.full_like(x, t), reduction="sum")
            return bce(real_b, 1.0) / Bg + bce(real_s, 1.0) / Sg + bce(fake_b, 0.0) / Bg + bce(fake_s, 0.0) / Sg
        if gl == "LSGAN":
            sq = lambda x, t: ((x - t) ** 2).sum()
            return 0.5 * (sq(real_b, 1.0) / Bg + sq(real_s, 1.0) / Sg + sq(fake_b, 0.0) / Bg + sq(fake_s, 0.0) / Sg)
        return (F.relu(1.0 - real_b).sum() / Bg + F.relu(1.0 - real_s).sum() / Sg
                + F.relu(fake_b + 1.0).sum() / Bg + F.relu(fake_s + 1.0).sum() / Sg)          # hinge

    def _critic_weights(self, B, S, Bg, Sg, dev):
        # a stream of batches brings a handful of (B, S) pairs over and over: kept per pair, and made on the device
        # (two fills each: a host-made vector would cost a blocking copy in the middle of the iteration)
        key = (B, S, Bg, Sg)
        if self._cw_cache is None:
            self._cw_cache = {}
        hit = self._cw_cache.get(key)
        if hit is None:
            if len(self._cw_cache) >= 256:
                self._cw_cache.clear()
            wb = torch.full((2 * B,), 1.0 / Bg, dtype=torch.float32, device=dev)
            wb[:B] = -1.0 / Bg
            ws = torch.full((2 * S,), 1.0 / Sg, dtype=torch.float32, device=dev)
            ws[:S] = -1.0 / Sg
            hit = self._cw_cache[key] = (wb, ws)
        return hit

    def _g_adversarial(self, val_b, val_s, Bg, Sg):
        gl = self.gan_loss
        if gl in ("WGANGP", "hinge"):
            return -val_b.sum() / Bg - val_s.sum() / Sg
        if gl == "NSGAN":
            bce = lambda x: F.binary_cross_entropy(x, torch.ones_like(x), reduction="sum")
            return bce(val_b) / Bg + bce(val_s) / Sg
        return ((val_b - 1.0) ** 2).sum() / Bg + ((val_s - 1.0) ** 2).sum() / Sg               # LSGAN

    # ------------------------------------------------------------------ pieces of one iteration
    @staticmethod
    def _gp_sample(label, out, u):
        """The sample at which the penalty is taken (reference losses.py:75-79)."""
        if u is None:
            u = torch.rand(label.shape[0], 1, device=out.device)
        mixed = (label * u + out * (1 - u)).detach().requires_grad_(True)
        nonzero = (mixed.max(dim=-1)[0] != 0).type(torch.float32).view(-1, 1)
        return softmax_to_hard(mixed, -1) * nonzero

    def _gradient_penalty(self, batch, label, out, u, Nv_g):
        """reference losses.py:73-89 with the voxel mean taken over the global voxel count."""
        sample = self._gp_sample(label, out, u)
        if getattr(self.discriminator, "piecewise_linear", False):
            with ops.input_grads_only():
                val_b, val_s = self.discriminator(batch, sample)
        else:
            val_b, val_s = self.discriminator(batch, sample)
        grad_b = torch.autograd.grad(val_b, sample, torch.ones_like(val_b), retain_graph=True, create_graph=True)[0]
        grad_s = torch.autograd.grad(val_s, sample, torch.ones_like(val_s), retain_graph=True, create_graph=True)[0]
        gp_b = ((grad_b.norm(2, 1) - 1) ** 2).sum() / Nv_g
        gp_s = ((grad_s.norm(2, 1) - 1) ** 2).sum() / Nv_g
        return gp_b + gp_s

    def _phase_d(self, batch, counts, do_d, program_z=None, voxel_z=None, noise=None, u=None, g_update=True):
        """train.py:80-99: generator forward, discriminator losses, d_loss.backward()."""
        G, D = self.generator, self.discriminator
        dev = self.device
        Bg, Sg, Nvg = counts
        stats = {}
        plan = batch.__dict__.get("_bg_plan") if hasattr(batch, "__dict__") else None
        if plan is not None:
            plan.begin_iteration()
        for p in self.opt_d.params:
            p.requires_grad = True
        self.opt_d.detach_grads()
        if program_z is None:
            program_z = torch.rand(batch.program_class_feature.shape[0], self.noise_dim, device=dev)
        if voxel_z is None and getattr(getattr(G, "voxelGNN", None), "noise_dim", 1) != 0:
            # (the reference draws it every iteration, train.py:86, and its generator never reads it: models.py:239)
            voxel_z = torch.rand(batch.voxel_feature.shape[0], self.noise_dim, device=dev)
        fused_ok = do_d and hasattr(D, "critic_terms") and (self.gp_lambda == 0 or getattr(D, "piecewise_linear", False))
        share = D.shared_coefficients() if (do_d and hasattr(D, "shared_coefficients")) else contextlib.nullcontext()
        with share:
            side = self._overlap_stream if fused_ok else None
            if side is not None:
                # what the critic needs from the graph and the parameters alone (operator powers,
                # coefficient table: chains of small kernels) runs beside the generator forward
                position_terms(batch)                      # shared with the generator: not on the side stream
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    D.prepare(batch)
            # the generator's autograd graph is only needed when its own update follows (train.py:106-113)
            if g_update:
                gen = G(batch, program_z, voxel_z, noise)
            else:
                with torch.no_grad():
                    derived = self._derived()
                    gen = G(batch, program_z, voxel_z, noise, derived) if derived is not None else \
                        G(batch, program_z, voxel_z, noise)
            if side is not None:
                cur.wait_stream(side)
            out, soft_out = gen[0], gen[1]
            if do_d:
                gp = None
                fused = None
                head = None
                manual_done = False
                if fused_ok:
                    sample = None
                    if self.gp_lambda != 0:      # labels are data in the fused pass: the sample point in one launch
                        sample = ops.gp_sample_raw(batch.voxel_label, soft_out.detach(),
                                                   u if u is not None else torch.rand(out.shape[0], device=dev))
                    plan_now = batch.__dict__.get("_bg_plan") if hasattr(batch, "__dict__") else None      # built by G's forward
                    if sample is not None and self.gan_loss == "WGANGP" and hasattr(D, "critic_objective") and plan_now is not None:
                        # the whole update below the coefficient table as one host call each way
                        wb, ws = self._critic_weights(plan_now.B, plan_now.S, Bg, Sg, dev)
                        head = None
                        if self.manual_critic:       # forward and backward right here, gradients assigned: no autograd pass
                            head = D.critic_objective(batch, batch.voxel_label, out.detach(), sample, wb, ws,
                                                      self.gp_lambda / Nvg, manual=True)
                            manual_done = head is not None
                        if head is None:
                            head = D.critic_objective(batch, batch.voxel_label, out.detach(), sample, wb, ws, self.gp_lambda / Nvg)
                    if head is None:
                        fused = D.critic_terms(batch, batch.voxel_label, out.detach(), sample, stacked=True)
                if head is not None:
                    pass
                elif fused is not None:
                    val_b_all, val_s_all, grads = fused
                    if grads is not None and self.gan_loss != "WGANGP":   # both penalties at once: grads[:, 0] and grads[:, 1]
                        gp = ((grads.norm(2, 2) - 1) ** 2).sum() / Nvg
                else:
                    fake_b, fake_s = D(detach_batch(batch), out.detach())
                    if self.gp_lambda != 0:
                        gp = self._gradient_penalty(detach_batch(batch), batch.voxel_label.detach(),
                                                    soft_out.detach(), u, Nvg)
                    real_b, real_s = D(batch, batch.voxel_label)
        if do_d:
            if head is not None:
                d_loss, pen = head
                stats["gp"] = pen / Nvg
            elif fused is not None and self.gan_loss == "WGANGP":
                # -E[real] + E[fake] over graphs and over stories as two dot products with +-1/count weights
                # and the penalty lambda/Nv * sum (||grad|| - 1)^2 over both outputs' gradients: ONE node
                wb, ws = self._critic_weights(val_b_all.shape[0] // 2, val_s_all.shape[0] // 2, Bg, Sg, dev)
                d_loss, pen = ops.CriticLoss.apply(val_b_all, val_s_all, grads, wb, ws, self.gp_lambda / Nvg)
                if grads is not None:
                    stats["gp"] = pen / Nvg
            else:
                if fused is not None:
                    (real_b, fake_b), (real_s, fake_s) = val_b_all.chunk(2, dim=0), val_s_all.chunk(2, dim=0)
                d_loss = self._d_adversarial(real_b, real_s, fake_b, fake_s, Bg, Sg)
            if gp is not None:
                d_loss = d_loss + self.gp_lambda * gp
                stats["gp"] = gp.detach()
            if not manual_done:
                d_loss.backward()
            self.opt_d.collect()
            stats["d_loss"] = d_loss.detach()
        return gen, stats

    def _phase_g(self, batch, counts, gen):
        """train.py:106-113: D frozen, generator loss through D, g_loss.backward()."""
        Bg, Sg, _ = counts
        out, soft_out, mask, att, _ = gen
        for p in self.opt_d.params:
            p.requires_grad = False
        self.opt_g.detach_grads()
        g_loss = None
        plan = batch.__dict__.get("_bg_plan") if hasattr(batch, "__dict__") else None
        if self.gan_loss in ("WGANGP", "hinge") and plan is not None and hasattr(self.discriminator, "generator_objective"):
            # -E[D_building] - E[D_story] through the frozen discriminator: one host call each way
            wb, ws = self._critic_weights(plan.B, plan.S, Bg, Sg, out.device)      # first halves: -1/count
            g_loss = self.discriminator.generator_objective(batch, out, wb, ws)
        if g_loss is None:
            val_b, val_s = self.discriminator(batch, out)
            g_loss = self._g_adversarial(val_b, val_s, Bg, Sg)
        stats = {"g_adv": g_loss.detach()}
        # get_program_ratio runs in every G step although its losses have zero weight by default (losses.py:65)
        # (no gradient is asked of it then: values only, two launches)
        with torch.set_grad_enabled(self.tr_weight != 0 or self.far_weight != 0):
            ratio, _, far = self.program_ratio_fn(batch, att["hard"], mask["hard"], 6)
        # smooth_l1_loss means over the 6 B class slots / the B graphs (losses.py:66-67), as local sum / global count
        if self.tr_weight != 0:
            tr = self.tr_weight * F.smooth_l1_loss(ratio.flatten(), batch.program_target_ratio, reduction="sum") / \
                (Bg * (ratio.numel() // max(int(batch.FAR.shape[0]), 1)))
            g_loss, stats["tr_loss"] = g_loss + tr, tr.detach()
        if self.far_weight != 0:
            fl = self.far_weight * F.smooth_l1_loss(far.view(far.shape[0]), batch.FAR, reduction="sum") / Bg
            g_loss, stats["far_loss"] = g_loss + fl, fl.detach()
        g_loss.backward()
        self.opt_g.collect()
        stats["g_loss"] = g_loss.detach()
        return stats

    def _reduce_max(self):
        """The collective that makes a per-rank measurement a group-wide decision (None with one rank)."""
        if not _collective(self.group):
            return None

        def reduce_max(t):
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                torch.cuda.current_stream().synchronize()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return t
        return reduce_max

    def prefetch(self, batch):
        """What the NEXT iteration needs from its batch alone -- the index plan and, with several ranks, the global
        counts -- issued now beside the running iteration (side stream; see plan.prefetch_plan)."""
        from .plan import prefetch_plan
        prefetch_plan(batch)
        prefetch_counts(batch, self.group)

    # ------------------------------------------------------------------ eager iteration
    def train_iteration(self, batch, i: int, program_z=None, voxel_z=None, noise: Optional[Sequence] = None,
                        u: Optional[torch.Tensor] = None):
        """Iteration ``i`` of an epoch (reference train.py:74-114).  Returns a dict of detached losses."""
        if hasattr(batch, "__dict__") and "_bg_plan_pending" in batch.__dict__:
            from .plan import build_plan
            build_plan(batch)               # started by prefetch(): read its (finished) flags, build the views
        counts = global_counts(batch, self.group)
        do_d = i % self.n_critic_d == 0
        if (self.collapse_check_every and i % self.collapse_check_every == 0 and hasattr(self.discriminator, "guard_collapse")
                and batch.voxel_feature.is_cuda and not torch.cuda.is_current_stream_capturing()):
            self.discriminator.guard_collapse(batch, reduce_max=self._reduce_max())
        gen, stats = self._phase_d(batch, counts, do_d, program_z, voxel_z, noise, u, g_update=i % self.n_critic_g == 0)
        if do_d:
            self.opt_d.all_reduce(self.group, self.grad_exchange)
            self.opt_d.step()
        if i % self.n_critic_g == 0:
            stats.update(self._phase_g(batch, counts, gen))
            self.opt_g.all_reduce(self.group, self.grad_exchange)
            self.opt_g.step()
            self.refresh_derived()
        else:
            for p in self.opt_d.params:
                p.requires_grad = False
        return stats

    # ------------------------------------------------------------------ epochs and checkpoints
    def save_checkpoint(self, path: str):
        """The generator's state_dict as the reference writes it (train.py:140-141): plain tensors under the
        reference's key names (the parameters here are views of one flat buffer; each is copied out so the file does
        not carry the whole buffer once per key), loadable by inference.load_generator / the reference's
        inference.py:74-76."""
        torch.save({k: v.detach().clone().cpu() for k, v in self.generator.state_dict().items()}, path)

    @staticmethod
    def settle_host():
        """Before a long run of iterations: one full collection, then ``gc.freeze()`` -- the networks, plans' code, the
        library bindings leave the cyclic collector's generations, so the collections the per-iteration autograd graphs
        trigger no longer walk them (the collector stays ON; measured 3-4 % of an eager step, tools/gc_probe.py).
        Process-wide, as gc.freeze() is; ``gc.unfreeze()`` undoes it."""
        import gc
        gc.collect()
        gc.freeze()

    def fit(self, loader, n_epochs: int, model_dir: Optional[str] = None, eval_period: int = 20, on_epoch=None):
        """The reference's epoch loop (train.py:73-75,133-141): every batch of ``loader`` (any iterable of batches,
        re-iterated per epoch) goes to the device and through one iteration whose index restarts with the epoch;
        every ``eval_period``-th epoch the generator is written to ``model_dir/<epoch>.ckpt`` and
        ``on_epoch(epoch, stats)`` (evaluation, logging) is called.  Returns the last iteration's losses."""
        import os
        stats = {}
        from .plan import prefetch_plan
        dev = torch.device(self.device)
        self.settle_host()

        def fetch(it):
            b = next(it, None)
            if b is not None and b.voxel_feature.device != dev:
                b = b.to(dev)
            return b
        for epoch in range(n_epochs):
            it = iter(loader)
            batch, i = fetch(it), 0
            while batch is not None:
                nxt = fetch(it)
                if nxt is not None and dev.type == "cuda":
                    self.prefetch(nxt)            # the next batch's index plan (and global counts): started before this
                stats = self.train_iteration(batch, i)      # iteration is issued, read when the next one begins
                batch, i = nxt, i + 1
            if epoch % eval_period == 0:
                if model_dir is not None:
                    os.makedirs(model_dir, exist_ok=True)
                    self.save_checkpoint(os.path.join(model_dir, "{}.ckpt".format(epoch)))
                if on_epoch is not None:
                    on_epoch(epoch, stats)
        return stats

    # ------------------------------------------------------------------ hipGraph iteration
    def capture(self, batch, warmup: int = 2, overlap: bool = False):
        """Capture the iteration for THIS batch (fixed shapes) into hipGraphs.

        Segments: A = phase D (G forward ... d_loss.backward); B = Adam(D) + phase G; C = Adam(G);
        A2/B2 = the same without a generator update.  Collectives run eagerly between segments.
        Parameters and optimizer state are restored afterwards, so capturing does not train.
        """
        if self.n_critic_d != 1:
            raise NotImplementedError("graph capture assumes n_critic_d == 1 (the reference default)")
        if warmup < 1:
            raise ValueError("capture needs >= 1 eager warm-up iteration (host-side caches such as the critic's "
                             "+-1/count weights are uploaded there, not inside a stream capture)")
        import gc
        gc.collect()        # autograd graphs of earlier eager iterations that only a reference cycle keeps alive hold
                            # AccumulateGrad nodes made on the default stream, which must not be pulled into the capture
        saved = (self.opt_g.state(), self.opt_d.state())
        counts = global_counts(batch, self.group)
        self._overlap_stream = torch.cuda.Stream() if overlap else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):
                self.train_iteration(batch, i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs = {}
        pool = None
        # With a process group the collectives' watchdog thread polls the events of finished work whenever it likes; in
        # the default ("global") capture mode such a query from ANOTHER thread during a capture here is an error that
        # aborts the process (seen once the host stopped blocking in prefetch_counts).  The collectives themselves run
        # between the segments, never inside one, so only this thread's calls need policing.
        mode = "thread_local" if _collective(self.group) else "global"

        def cap(fn):
            nonlocal pool
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                res = fn()
            pool = g.pool()
            return g, res

        # iteration WITH generator update
        gA, (gen, statsA) = cap(lambda: self._phase_d(batch, counts, True))

        def seg_b():
            self.opt_d.step()
            return self._phase_g(batch, counts, gen)
        gB, statsB = cap(seg_b)
        self._derived()                       # the buffers must exist before the graphs that read / refresh them

        def seg_c():
            self.opt_g.step()
            self.refresh_derived()
        gC, _ = cap(seg_c)
        graphs["with_g"] = (gA, gB, gC, {**statsA, **statsB})
        # iteration WITHOUT generator update
        gA2, (gen2, statsA2) = cap(lambda: self._phase_d(batch, counts, True, g_update=False))
        gB2, _ = cap(lambda: self.opt_d.step())
        graphs["no_g"] = (gA2, gB2, statsA2)
        self._graphs = graphs
        self._captured_batch = batch
        self._keep = (gen, gen2)
        self._overlap_stream = None    # eager iterations stay on one stream
        self.opt_g.restore(saved[0])
        self.opt_d.restore(saved[1])
        self.refresh_derived()
        torch.cuda.synchronize()

    def train_iteration_graphed(self, i: int):
        """Replay the captured iteration.  A replay runs the discriminator in the form it had when it was captured
        (collapsed polynomial or layer stack); every ``collapse_check_every`` replays the guard is measured again on
        the captured batch (group-wide, one host read) and a changed decision re-captures the graphs."""
        if self._graphs is None:
            raise RuntimeError("call capture(batch) first")
        if (self.collapse_check_every and i > 0 and i % self.collapse_check_every == 0
                and hasattr(self.discriminator, "guard_collapse") and self._captured_batch is not None):
            gnn = getattr(self.discriminator, "voxelGNN_D", None)
            before = getattr(gnn, "collapse_ok", None)
            self.discriminator.guard_collapse(self._captured_batch, reduce_max=self._reduce_max())
            if getattr(gnn, "collapse_ok", None) != before:
                self.capture(self._captured_batch, warmup=1)
        if i % self.n_critic_g == 0:
            gA, gB, gC, stats = self._graphs["with_g"]
            gA.replay()
            self.opt_d.all_reduce(self.group, self.grad_exchange)
            gB.replay()
            self.opt_g.all_reduce(self.group, self.grad_exchange)
            gC.replay()
        else:
            gA, gB, stats = self._graphs["no_g"]
            gA.replay()
            self.opt_d.all_reduce(self.group, self.grad_exchange)
            gB.replay()
        return stats
