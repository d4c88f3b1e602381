"""End-to-end parity of the HIP-backed networks against the CPU oracle on identical weights,
inputs and noise: generator forward, discriminator forward, first-order gradients, the WGAN-GP
double backward, and one full training iteration.  fp32 tolerance: rtol 1e-4 on activations
(north_star), a looser bound on gradients that pass through 12-16 layers.
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu



@pytest.fixture(params=[1, 0], ids=["chebyshev", "monomial"])
def d_basis(request):
    """Run a test under both bases of the collapsed discriminator (ops.D_BASIS: T_k(A), the default, and A^k)."""
    from building_gan_amd import ops
    old = ops.D_BASIS
    ops.D_BASIS = request.param
    yield request.param
    ops.D_BASIS = old


def _setup(preset, seed, ragged, dev, layers=(4, 12)):
    from building_gan_amd import models as hm
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from oracle import networks as om
    cpu = make_batch(preset, seed=seed, ragged=ragged)
    sz = batch_sizes(cpu)
    torch.manual_seed(seed)
    Go = om.Generator(7, 7, 128, 32, layers[0], layers[1])
    Do = om.DiscriminatorVoxel(7, 6, 128, layers[1])
    Gh = hm.Generator(7, 7, 128, 32, layers[0], layers[1], dev).to(dev)
    Dh = hm.DiscriminatorVoxel(7, 6, 128, layers[1]).to(dev)
    Gh.load_state_dict(Go.state_dict())
    Dh.load_state_dict(Do.state_dict())
    n_att = 1 + sum(1 for i in range(layers[1]) if i % 2 == 0 and i != 0)
    noise = [(torch.empty(sz["Nv"], 2).exponential_(), torch.empty(sz["Ec"]).exponential_()) for _ in range(n_att)]
    pz, vz = torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32)
    gpu = copy.deepcopy(cpu).to(dev)
    return cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz


def _to(noise, dev):
    return [(a.to(dev), b.to(dev)) for a, b in noise]


def _assert_grads_close(named_ref, named_hip, bar, what):
    """max |hip - ref| of every parameter gradient against max(the tensor's own largest gradient, 1e-3 of the
    network's largest): additive terms of the linear critic have an exactly-zero gradient in exact arithmetic
    (they shift D(real) and D(fake) alike), which is rounding noise on both sides.  Returns the worst ratio."""
    refs = [(n, p.grad) for n, p in named_ref]
    top = max(float(g.abs().max()) for _, g in refs if g is not None)
    worst = 0.0
    for (n, g), (_, ph) in zip(refs, named_hip):
        if g is None:
            assert ph.grad is None or float(ph.grad.abs().max()) <= 1e-6 * top, (what, n)
            continue
        hg = ph.grad.cpu() if ph.grad is not None else torch.zeros_like(g)
        e = float((hg - g).abs().max()) / max(float(g.abs().max()), 1e-3 * top)
        assert e <= bar, (what, n, e)
        worst = max(worst, e)
    return worst


def _assert_adam_steps_agree(named_ref, named_hip, before, lr=1e-4):
    """After ONE Adam step from identical weights: the first step moves every element by -lr * g / (|g| + eps),
    i.e. by lr against the gradient's sign, so wherever the reference gradient is not rounding noise (>= 1 % of
    its tensor's max) both sides must have moved the same way, to 5 % of lr."""
    checked = 0
    named_ref = list(named_ref)
    top = max(float(p.grad.abs().max()) for _, p in named_ref if p.grad is not None)
    for (n, po), (_, ph) in zip(named_ref, named_hip):
        g = po.grad
        if g is None or float(g.abs().max()) < 1e-3 * top:     # rounding noise around an exactly-zero gradient
            continue
        sig = g.abs() >= 1e-2 * g.abs().max()
        do, dh = po.detach() - before[n], ph.detach().cpu() - before[n]
        assert float(do[sig].abs().min()) >= 0.5 * lr, n      # the oracle did step (|g| may be within ~10x of eps)
        bad = ((do - dh).abs() > 0.05 * lr) & sig
        assert int(bad.sum()) == 0, (n, int(bad.sum()), int(sig.sum()))
        checked += int(sig.sum())
    assert checked > 0


def test_state_dict_keys_match_reference_checkpoint_layout(cuda):
    from building_gan_amd import models as hm
    G = hm.Generator(7, 7, 128, 32, 4, 12, cuda)
    keys = set(G.state_dict().keys())
    expect = {"programGNN.enc.0.weight", "programGNN.block.messageMLP.0.weight", "programGNN.block.updateMLP.0.bias",
              "voxelGNN.pe", "voxelGNN.enc.0.weight", "voxelGNN.attention.theta", "voxelGNN.attention.dec.1.weight",
              "voxelGNN.attention.NN_v.weight", "voxelGNN.attention.NN_p.bias", "voxelGNN.block.messageMLP.0.weight",
              "voxelGNN.block.updateMLP.0.weight"}
    assert expect <= keys
    assert sum(p.numel() for p in G.parameters()) == 228418
    D = hm.DiscriminatorVoxel(7, 6, 128, 12)
    assert sum(p.numel() for p in D.parameters()) == 331010


@pytest.mark.parametrize("preset,ragged", [("tiny", False), ("small", True)])
def test_generator_forward_parity(cuda, preset, ragged):
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, 11, ragged, cuda)
    with torch.no_grad():
        ro = Go(cpu, pz, vz, noise)
        rh = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))
    out_o, soft_o, mask_o, att_o, idx_o = ro
    out_h, soft_h, mask_h, att_h, idx_h = rh
    torch.testing.assert_close(mask_h["soft"].cpu(), mask_o["soft"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(att_h["soft"].cpu(), att_o["soft"], rtol=1e-4, atol=1e-5)
    # discrete outputs: identical unless a near-tie flips (none at these sizes / seeds)
    assert torch.equal(idx_h.cpu(), idx_o)
    torch.testing.assert_close(out_h.cpu(), out_o, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(soft_h.cpu(), soft_o, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mask_h["hard"].cpu(), mask_o["hard"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("preset,ragged", [("tiny", False), ("small", True)])
def test_discriminator_forward_backward_parity(cuda, preset, ragged, d_basis):
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, 12, ragged, cuda)
    lab_o = cpu.voxel_label.clone().requires_grad_(True)
    lab_h = gpu.voxel_label.clone().requires_grad_(True)
    bo, so = Do(cpu, lab_o)
    bh, sh = Dh(gpu, lab_h)
    torch.testing.assert_close(bh.cpu(), bo, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(sh.cpu(), so, rtol=1e-4, atol=1e-4)
    (bo.sum() + 2 * so.sum()).backward()
    (bh.sum() + 2 * sh.sum()).backward()
    assert float((lab_h.grad.cpu() - lab_o.grad).abs().max()) <= 1e-4 * float(lab_o.grad.abs().max())
    # measured 3e-5 (fp32 oracle vs HIP, the oracle itself ~1e-5 from float64: test_gpu_sizes.py)
    _assert_grads_close(Do.named_parameters(), Dh.named_parameters(), 2e-4, "D")


def test_gradient_penalty_double_backward_parity(cuda, d_basis):
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.losses import compute_gradient_penalty as gp_h
    from oracle.objectives import compute_gradient_penalty as gp_o
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 13, True, cuda)
    torch.manual_seed(13)
    u = torch.rand(sz["Nv"], 1)
    soft = torch.rand(sz["Nv"], 6).softmax(-1) * (torch.rand(sz["Nv"], 1) > 0.3).float()
    go, _, _ = gp_o(Do, detach_batch(cpu), cpu.voxel_label, soft, u)
    gh, _, _ = gp_h(Dh, detach_batch(gpu), gpu.voxel_label, soft.to(cuda), u.to(cuda))
    torch.testing.assert_close(gh.cpu(), go, rtol=1e-5, atol=1e-7)
    go.backward()
    gh.backward()
    # measured 4e-7 .. 3e-6 against float64 (test_gpu_sizes.py): north_star's 1e-4 with room to spare
    _assert_grads_close(Do.named_parameters(), Dh.named_parameters(), 1e-4, "penalty")


def test_generator_backward_through_discriminator(cuda, d_basis):
    from oracle.objectives import g_loss_fn
    from building_gan_amd.losses import VolumetricDesignLoss_G
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 14, True, cuda)
    out_o, soft_o, mask_o, att_o, _ = Go(cpu, pz, vz, noise)
    lo = g_loss_fn(Do(cpu, out_o), cpu, att_o["hard"], mask_o["hard"], 6)[0]
    lo.backward()
    out_h, soft_h, mask_h, att_h, _ = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))
    lh = VolumetricDesignLoss_G()(Dh(gpu, out_h), gpu, att_h["hard"], mask_h["hard"], 6)[0]
    lh.backward()
    torch.testing.assert_close(lh.cpu(), lo, rtol=1e-4, atol=1e-4)
    # measured 3e-6 (HIP) / 1e-5 (fp32 oracle) against float64 at this size (test_gpu_sizes.py)
    _assert_grads_close(Go.named_parameters(), Gh.named_parameters(), 1e-4, "G")


@pytest.mark.parametrize("preset,ragged", [("small", True), ("dataset", "jitter")])
def test_program_ratio_parity(cuda, preset, ragged):
    """get_program_ratio in both of its forms -- values only (bg_program_ratio_f32, two launches: what evaluation and a
    generator step with zero ratio weights call) and op by op under autograd -- against the oracle and against each other."""
    from building_gan_amd.util_graph import get_program_ratio as h
    from oracle.pointer import get_program_ratio as o
    cpu, gpu, sz, *_ = _setup(preset, 15, ragged, cuda)
    torch.manual_seed(15)
    att = torch.rand(sz["Ec"], 1)
    mask = (torch.rand(sz["Nv"], 1) > 0.4).float()
    ro = o(cpu, att, mask, 6)
    rh = h(gpu, att.to(cuda), mask.to(cuda), 6)                     # no gradient asked for: the two-launch form
    att_g = att.to(cuda).requires_grad_()
    rg = h(gpu, att_g, mask.to(cuda), 6)                            # op by op, differentiable
    assert rg[0].requires_grad and not rh[0].requires_grad
    with torch.no_grad():
        rn = h(gpu, att_g, mask.to(cuda), 6)                        # values only again; the same bits as the first call
    for a, b, c, d in zip(rh, ro, rg, rn):
        assert a.shape == b.shape == c.shape
        torch.testing.assert_close(a.cpu(), b, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(c.detach().cpu(), b, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(a, c.detach(), rtol=2e-6, atol=1e-7)          # the forms differ in summation order only
        assert torch.equal(a, d)


def test_trainer_matches_oracle_iterations(cuda):
    """Two full iterations (i=0 with the generator update, i=1 without) against the CPU oracle."""
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.train import Trainer
    from oracle.objectives import train_iteration
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 21, True, cuda)
    tr = Trainer(cuda)
    tr.generator.load_state_dict(Go.state_dict())
    tr.discriminator.load_state_dict(Do.state_dict())
    og = torch.optim.Adam(Go.parameters(), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(Do.parameters(), lr=1e-4, betas=(0.5, 0.999))
    before = {n: p.detach().clone() for n, p in list(Do.named_parameters()) + list(Go.named_parameters())}
    torch.manual_seed(21)
    for i in (0, 1):
        u = torch.rand(sz["Nv"], 1)
        so = train_iteration(Go, Do, og, od, cpu, detach_batch, i, pz, vz, noise, u)
        sh = tr.train_iteration(gpu, i, pz.to(cuda), vz.to(cuda), _to(noise, cuda), u.to(cuda))
        # iteration 1 runs on weights that already took one (sign-like) Adam step on each side
        tol = 2e-3 if i == 0 else 2e-2
        # d_loss = -E[real] + E[fake] + 10 gp is a difference of terms several times its own size
        torch.testing.assert_close(sh["d_loss"].cpu(), so["d_loss"], rtol=2.5 * tol, atol=5e-3)
        torch.testing.assert_close(sh["gp"].cpu(), so["gp"], rtol=tol, atol=1e-4)
        if i == 0:
            # g_loss is evaluated AFTER the discriminator's first Adam step (a +-lr move per weight whose
            # sign is rounding-sensitive where the gradient is ~0), so it only agrees to ~1e-2
            torch.testing.assert_close(sh["g_loss"].cpu().view(-1), so["g_loss"].view(-1), rtol=1e-2, atol=5e-3)
            # both networks have taken exactly one Adam step and the oracle still holds the gradients it stepped on
            # (the gradients themselves: test_gpu_sizes.py::test_trainer_flat_gradients_match_oracle)
            _assert_adam_steps_agree(Do.named_parameters(), tr.discriminator.named_parameters(), before)
            _assert_adam_steps_agree(Go.named_parameters(), tr.generator.named_parameters(), before)


def test_graph_capture_replays_and_restores_state(cuda):
    from building_gan_amd.synthetic import make_batch
    from building_gan_amd.train import Trainer
    batch = make_batch("small", seed=3, ragged=True).to(cuda)
    torch.manual_seed(0)
    tr = Trainer(cuda)
    before_g, before_d = tr.opt_g.flat.clone(), tr.opt_d.flat.clone()
    tr.capture(batch)
    assert torch.equal(tr.opt_g.flat, before_g) and torch.equal(tr.opt_d.flat, before_d)
    assert int(tr.opt_d.t.item()) == 0
    for i in range(6):
        stats = tr.train_iteration_graphed(i)
    torch.cuda.synchronize()
    assert int(tr.opt_d.t.item()) == 6 and int(tr.opt_g.t.item()) == 2
    assert torch.isfinite(stats["d_loss"]).all() and torch.isfinite(tr.opt_d.flat).all()
    assert not torch.equal(tr.opt_g.flat, before_g) and not torch.equal(tr.opt_d.flat, before_d)
    # the composed generator weights kept for critic-only iterations follow every generator update
    with torch.no_grad():
        fresh = tr.generator.voxelGNN.derived_weights()
    assert tr._g_derived is not None
    for kept, now in zip(tr._g_derived, fresh):
        assert torch.equal(kept, now)
    # an eager iteration on the same trainer still works after replays
    s2 = tr.train_iteration(batch, 7)
    assert torch.isfinite(s2["d_loss"]).all()
    tr.train_iteration(batch, 10)                                   # eager generator update: refreshed in place
    with torch.no_grad():
        fresh = tr.generator.voxelGNN.derived_weights()
    for kept, now in zip(tr._g_derived, fresh):
        assert torch.equal(kept, now)
    tr.generator.load_state_dict({k: v * 0.5 for k, v in tr.generator.state_dict().items()})
    with torch.no_grad():
        fresh = tr.generator.voxelGNN.derived_weights()
    for kept, now in zip(tr._g_derived, fresh):
        assert torch.equal(kept, now)


def test_graph_replays_stay_finite_and_do_not_depend_on_work_queued_between_segments(cuda):
    """Regression (round 3).  The captured iteration used to contain hipMemsetAsync / hipMemcpy2DAsync nodes (the
    coefficient recurrence's zero fills); replayed after a synchronisation and ANY eager kernel on the stream, the
    table came out zeroed -- g_adv 0.068 instead of -20.99 -- and the second replay of the generator-update graph
    went NaN.  Fills and copies are kernels of the library now (bg_zero_bytes / bg_copy2d_bytes)."""
    from building_gan_amd.synthetic import make_batch
    from building_gan_amd.train import Trainer

    def run(disturb, n_iter):
        batch = make_batch("small", seed=3, ragged=True).to(cuda)
        torch.manual_seed(0)
        tr = Trainer(cuda)
        tr.capture(batch)
        seen = []
        for i in range(n_iter):
            if i % tr.n_critic_g == 0:
                gA, gB, gC, stats = tr._graphs["with_g"]
                segs = (gA, gB, gC)
            else:
                gA, gB, stats = tr._graphs["no_g"]
                segs = (gA, gB)
            for g in segs:
                g.replay()
                if disturb:
                    torch.cuda.synchronize()
                    junk = [torch.full((1 + 37 * k,), float("nan"), device=cuda) for k in range(8)]
                    del junk
            torch.cuda.synchronize()
            seen.append({k: v.clone() for k, v in stats.items()})
        return seen, tr.opt_d.flat.clone(), tr.opt_g.flat.clone()

    plain, d0, g0 = run(False, 11)
    mixed, d1, g1 = run(True, 11)
    for i, (a, b) in enumerate(zip(plain, mixed)):
        for k in a:
            assert torch.isfinite(a[k]).all(), (i, k, a[k])
            assert torch.equal(a[k], b[k]), (i, k, a[k], b[k])
    assert torch.isfinite(d0).all() and torch.isfinite(g0).all()
    assert torch.equal(d0, d1) and torch.equal(g0, g1)


def test_isolated_voxel_falls_back_to_two_linear_path(cuda):
    """A 1x1x1 building has a voxel without in-edges: mean aggregation gives 0 there (not W x_i + b),
    so the composed-weight shortcut is illegal and the literal two-Linear path must be taken."""
    import numpy as np
    from building_gan_amd import models as hm
    from building_gan_amd.graph import Batch
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import batch_sizes, make_graph
    from oracle import networks as om
    rng = np.random.default_rng(5)
    data = [make_graph(rng, 2, 3, 2, [2, 3], data_id=0), make_graph(rng, 1, 1, 1, [2], data_id=1)]
    cpu = Batch.from_data_list(data)
    sz = batch_sizes(cpu)
    gpu = copy.deepcopy(cpu).to(cuda)
    assert not build_plan(gpu).vox_all_has
    torch.manual_seed(3)
    Go, Do = om.Generator(7, 7, 128, 32, 2, 4), om.DiscriminatorVoxel(7, 6, 128, 4)
    Gh, Dh = hm.Generator(7, 7, 128, 32, 2, 4, cuda).to(cuda), hm.DiscriminatorVoxel(7, 6, 128, 4).to(cuda)
    Gh.load_state_dict(Go.state_dict())
    Dh.load_state_dict(Do.state_dict())
    noise = [(torch.empty(sz["Nv"], 2).exponential_(), torch.empty(sz["Ec"]).exponential_()) for _ in range(2)]
    pz, vz = torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32)
    out_o, soft_o, _, _, idx_o = Go(cpu, pz, vz, noise)
    out_h, soft_h, _, _, idx_h = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))
    assert torch.equal(idx_h.cpu(), idx_o)
    torch.testing.assert_close(soft_h.detach().cpu(), soft_o.detach(), rtol=1e-4, atol=1e-5)
    lab_o = cpu.voxel_label.clone().requires_grad_(True)
    lab_h = gpu.voxel_label.clone().requires_grad_(True)
    bo, so = Do(cpu, lab_o)
    bh, sh = Dh(gpu, lab_h)
    torch.testing.assert_close(bh.detach().cpu(), bo.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(sh.detach().cpu(), so.detach(), rtol=1e-4, atol=1e-4)
    (bo.sum() + so.sum()).backward()
    (bh.sum() + sh.sum()).backward()
    torch.testing.assert_close(lab_h.grad.cpu(), lab_o.grad, rtol=1e-3, atol=1e-4)


def test_stack_node_equals_per_layer_nodes(cuda):
    """ops.AffineStack against the same layers run as separate autograd nodes: outputs, first-order
    gradients and the gradient-penalty double backward."""
    from building_gan_amd import models as hm
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.losses import compute_gradient_penalty
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 31, True, cuda)
    torch.manual_seed(31)
    u = torch.rand(sz["Nv"], 1, device=cuda)
    soft = (torch.rand(sz["Nv"], 6).softmax(-1) * (torch.rand(sz["Nv"], 1) > 0.3).float()).to(cuda)
    res = {}
    for flag in (True, False):
        hm.COMPOSE_STACK = flag
        try:
            Dh.zero_grad()
            b, s = Dh(gpu, gpu.voxel_label)
            gp, _, _ = compute_gradient_penalty(Dh, detach_batch(gpu), gpu.voxel_label, soft, u)
            (b.sum() - s.sum() + 10 * gp).backward()
            res[flag] = (b.detach().clone(), s.detach().clone(), gp.detach().clone(),
                         [p.grad.clone() for p in Dh.parameters()])
        finally:
            hm.COMPOSE_STACK = True
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[True][2], res[False][2], rtol=1e-4, atol=1e-6)
    for (n, _), a, c in zip(Dh.named_parameters(), res[True][3], res[False][3]):
        scale = float(c.abs().max()) + 1e-6
        assert float((a - c).abs().max()) <= 1e-3 * scale, n


def test_collapsed_discriminator_accuracy_against_fp64(cuda, d_basis):
    """The polynomial (collapsed) evaluation of the linear discriminator GNN and the layer-by-layer
    stack are two fp32 roundings of the same function: measure both against the oracle run in
    float64 on dataset-sized graphs and require the collapsed one to be within the fp32 tolerance."""
    from building_gan_amd import models as hm
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("dataset", 31, True, cuda)
    D64 = copy.deepcopy(Do).double()
    cpu64 = copy.deepcopy(cpu)
    cpu64.voxel_feature = cpu64.voxel_feature.double()
    lab64 = cpu.voxel_label.double().requires_grad_(True)
    b64, s64 = D64(cpu64, lab64)
    (b64.sum() + 2 * s64.sum()).backward()
    ref = {n: p.grad for n, p in D64.named_parameters()}
    errs = {}
    try:
        for name, flag in (("collapsed", True), ("stack", False)):
            hm.COLLAPSE_D = flag
            gpu.__dict__.pop("_bg_plan", None)
            Dh.zero_grad()
            lab = gpu.voxel_label.clone().requires_grad_(True)
            bh, sh = Dh(gpu, lab)
            (bh.sum() + 2 * sh.sum()).backward()
            e_out = max(float((bh.cpu().double() - b64).abs().max() / b64.abs().max()),
                        float((sh.cpu().double() - s64).abs().max() / s64.abs().max()))
            e_lab = float((lab.grad.cpu().double() - lab64.grad).abs().max() / lab64.grad.abs().max())
            e_par = max(float((p.grad.cpu().double() - ref[n]).abs().max() / (ref[n].abs().max() + 1e-12))
                        for n, p in Dh.named_parameters())
            errs[name] = (e_out, e_lab, e_par)
    finally:
        hm.COLLAPSE_D = True
    print("relative errors vs float64 (outputs, label gradient, parameter gradients):", errs)
    assert errs["collapsed"][0] <= 1e-4 and errs["collapsed"][1] <= 1e-3 and errs["collapsed"][2] <= 2e-3


def _highpass_block(D, t):
    """Blend the discriminator's block towards v <- v - A v (updateMLP = [0 | I], messageMLP = [0 | -I | 0]):
    v_L = (I - A)^L v_0 is a sum of binomially weighted operator powers that cancel on smooth inputs -- benign layer by
    layer, ill-conditioned as a polynomial."""
    blk = D.voxelGNN_D.block
    H2 = blk.updateMLP[0].weight.shape[0]
    eye = torch.eye(H2)
    with torch.no_grad():
        Wu, Wm = blk.updateMLP[0].weight, blk.messageMLP[0].weight
        Wu.mul_(1 - t).add_(t * torch.cat([torch.zeros(H2, H2), eye], 1).to(Wu.device))
        Wm.mul_(1 - t).add_(t * torch.cat([torch.zeros(H2, H2), -eye, torch.zeros(H2, 3)], 1).to(Wm.device))
        blk.updateMLP[0].bias.mul_(1 - t)
        blk.messageMLP[0].bias.mul_(1 - t)


def test_collapsed_discriminator_guard(cuda):
    """(i) the collapsed evaluation stays as accurate as the layer-by-layer stack when the
    block weights grow (x1 .. x8; both measured against the oracle in float64) and the guard leaves it on;
    (ii) on weights that make the polynomial cancel (a high-pass block) the guard's amplification estimate exceeds
    its limit, the discriminator switches to the stack, and its result stays within the fp32 tolerance where the
    forced collapsed form does not; (iii) the Trainer runs the guard and keeps training on the stack."""
    from building_gan_amd import models as hm
    from building_gan_amd.plan import build_plan
    from building_gan_amd.train import Trainer
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("dataset", 33, "jitter", cuda)
    cpu64 = copy.deepcopy(cpu)
    cpu64.voxel_feature = cpu64.voxel_feature.double()
    init = copy.deepcopy(Do.state_dict())

    def errors(Dref):
        D64 = copy.deepcopy(Dref).double()
        Dh.load_state_dict(Dref.state_dict())
        with torch.no_grad():
            v64 = D64.voxelGNN_D(cpu64.voxel_feature, cpu.voxel_label.double(), cpu.voxel_edge, cpu.voxel_feature_batch,
                                 cpu.voxel_floor_cluster, cpu.voxel_feature_batch)
            out = {}
            for name, flag in (("collapsed", True), ("stack", False)):
                hm.COLLAPSE_D = flag
                try:
                    gpu.__dict__.pop("_bg_plan", None)
                    Dh.voxelGNN_D.collapse_ok = True
                    v = Dh.voxelGNN_D(gpu.voxel_feature, gpu.voxel_label, build_plan(gpu)).cpu().double()
                finally:
                    hm.COLLAPSE_D = True
                out[name] = float(((v - v64).abs().amax(0) / v64.abs().amax(0)).max())
            amp = Dh.guard_collapse(gpu)
        return out, amp, v64

    for scale in (1, 2, 4, 8):
        Do.load_state_dict(init)
        with torch.no_grad():
            for p in Do.voxelGNN_D.block.parameters():
                p.mul_(scale)
        e, amp, _ = errors(Do)
        print("block weights x%d: amplification %.1f; error / column max vs float64: collapsed %.2e, stack %.2e" %
              (scale, amp, e["collapsed"], e["stack"]))
        assert Dh.voxelGNN_D.collapse_ok and amp <= Dh.COLLAPSE_AMPLIFICATION_LIMIT
        assert e["collapsed"] <= 2e-5 and e["collapsed"] <= 2 * e["stack"]          # measured 1.4e-6 .. 4.3e-6
    # (ii) a block that makes the polynomial cancel
    Do.load_state_dict(init)
    _highpass_block(Do, 1.0)
    e, amp, v64 = errors(Do)
    print("high-pass block: amplification %.3e (limit %.1f); error / column max vs float64: collapsed %.2e, stack %.2e" %
          (amp, Dh.COLLAPSE_AMPLIFICATION_LIMIT, e["collapsed"], e["stack"]))
    assert amp > Dh.COLLAPSE_AMPLIFICATION_LIMIT and not Dh.voxelGNN_D.collapse_ok
    assert e["stack"] <= 1e-4 and e["collapsed"] > e["stack"]
    with torch.no_grad():                                        # guarded: the default path now IS the stack
        v = Dh.voxelGNN_D(gpu.voxel_feature, gpu.voxel_label, build_plan(gpu)).cpu().double()
        assert float(((v - v64).abs().amax(0) / v64.abs().amax(0)).max()) <= 1e-4
        assert Dh.critic_terms(gpu, gpu.voxel_label, gpu.voxel_label) is None     # the fused pass steps aside too
    # back inside: the guard switches the collapsed form on again
    Do.load_state_dict(init)
    Dh.load_state_dict(init)
    Dh.guard_collapse(gpu)
    assert Dh.voxelGNN_D.collapse_ok
    # (iii) the trainer
    tr = Trainer(cuda, collapse_check_every=2)
    tr.discriminator.load_state_dict(Do.state_dict())
    s0 = tr.train_iteration(gpu, 1)
    assert tr.discriminator.voxelGNN_D.collapse_ok
    _highpass_block(tr.discriminator, 1.0)
    s1 = tr.train_iteration(gpu, 2)                              # i % 2 == 0: guard runs, stack takes over
    assert not tr.discriminator.voxelGNN_D.collapse_ok
    assert torch.isfinite(s1["d_loss"]).all() and torch.isfinite(tr.opt_d.grad).all()


def test_collapsed_discriminator_stays_conditioned_while_training(cuda):
    """The reason for the Chebyshev basis: train the synthetic stream for 1000 iterations, then on the
    TRAINED parameters (i) the guard's amplification in the basis T_k(A) is far below its limit and the collapsed form is
    still on, (ii) the collapsed GNN output equals the oracle run layer by layer in float64 to the fp32 tolerance of the
    untrained case, (iii) the same parameters in the monomial basis A^k amplify several times more (printed; on the
    trajectories measured in tools/basis_conditioning.py it passes the limit within ~1000 iterations)."""
    from building_gan_amd import ops
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import make_batch
    from building_gan_amd.train import Trainer
    from oracle import networks as om
    assert ops.D_BASIS == ops.CHAIN_CHEB                       # the default
    cpu = [make_batch("dataset", seed=k, ragged="jitter") for k in range(4)]
    gpu = [copy.deepcopy(b).to(cuda) for b in cpu]            # Batch.to moves in place
    torch.manual_seed(2)
    tr = Trainer(cuda, collapse_check_every=250)
    D = tr.discriminator
    for i in range(1000):
        b = gpu[i % 4]
        b.__dict__.pop("_bg_plan", None)
        tr.train_iteration(b, i)
    assert D.voxelGNN_D.collapse_ok
    amp_c = D.guard_collapse(gpu[0])
    try:
        ops.D_BASIS = ops.CHAIN_POWER
        gpu[0].__dict__.pop("_bg_plan", None)
        amp_m = D.guard_collapse(gpu[0])
    finally:
        ops.D_BASIS = ops.CHAIN_CHEB
        gpu[0].__dict__.pop("_bg_plan", None)
        D.guard_collapse(gpu[0])
    Do = om.DiscriminatorVoxel(7, 6, 128, 12)
    Do.load_state_dict({k: v.cpu() for k, v in D.state_dict().items()})
    D64 = Do.double()
    c = cpu[0]
    with torch.no_grad():
        v64 = D64.voxelGNN_D(c.voxel_feature.double(), c.voxel_label.double(), c.voxel_edge, c.voxel_feature_batch,
                             c.voxel_floor_cluster, c.voxel_feature_batch)
        v = D.voxelGNN_D(gpu[0].voxel_feature, gpu[0].voxel_label, build_plan(gpu[0])).cpu().double()
    err = float(((v - v64).abs().amax(0) / v64.abs().amax(0)).max())
    print("after 1000 iterations: amplification Chebyshev %.1f, monomial %.1f (limit %.1f); collapsed output vs float64 "
          "oracle, error / column max %.2e" % (amp_c, amp_m, D.COLLAPSE_AMPLIFICATION_LIMIT, err))
    assert D.voxelGNN_D.collapse_ok and amp_c <= D.COLLAPSE_AMPLIFICATION_LIMIT / 3
    assert err <= 2e-5
    assert amp_m >= 2 * amp_c


def test_fused_critic_terms_match_separate_passes_and_oracle_gradients(cuda, d_basis):
    """critic_terms (one pass for D(real), D(fake) and the closed-form penalty gradients) against the
    oracle's three separate passes with torch.autograd.grad (reference losses.py:83-84)."""
    from building_gan_amd.util import softmax_to_hard
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 21, True, cuda)
    torch.manual_seed(21)
    fake = torch.zeros(sz["Nv"], 6).scatter_(1, torch.randint(0, 6, (sz["Nv"], 1)), 1.0) * (torch.rand(sz["Nv"], 1) > 0.3)
    u = torch.rand(sz["Nv"], 1)
    mixed = cpu.voxel_label * u + fake * (1 - u)
    sample = (softmax_to_hard(mixed, -1) * (mixed.max(-1)[0] != 0).float().view(-1, 1)).detach()
    so = sample.clone().requires_grad_(True)
    rb, rs = Do(cpu, cpu.voxel_label)
    fb, fs = Do(cpu, fake)
    hb, hs = Do(cpu, so)
    gb = torch.autograd.grad(hb, so, torch.ones_like(hb), retain_graph=True, create_graph=True)[0]
    gs = torch.autograd.grad(hs, so, torch.ones_like(hs), retain_graph=True, create_graph=True)[0]
    res = Dh.critic_terms(gpu, gpu.voxel_label, fake.to(cuda), sample.to(cuda))
    assert res is not None
    for got, want in zip(res, (rb, rs, fb, fs, gb, gs)):
        torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=2e-5)
    # the penalty's parameter gradients (double backward in the oracle, first order here)
    pen_o = ((gb.norm(2, 1) - 1) ** 2).mean() + ((gs.norm(2, 1) - 1) ** 2).mean()
    pen_h = ((res[4].norm(2, 1) - 1) ** 2).mean() + ((res[5].norm(2, 1) - 1) ** 2).mean()
    torch.testing.assert_close(pen_h.cpu(), pen_o, rtol=1e-4, atol=1e-6)
    pen_o.backward()
    pen_h.backward()
    _assert_grads_close(Do.named_parameters(), Dh.named_parameters(), 1e-4, "penalty (closed form)")


@pytest.mark.parametrize("gan_loss", ["NSGAN", "LSGAN", "hinge"])
def test_trainer_other_objectives_match_oracle(cuda, gan_loss):
    """The non-default objectives of reference losses.py:14-35,52-61 in the trainer (NSGAN puts a sigmoid
    on the decoders, reference train.py:65; no gradient penalty): first iteration against the oracle."""
    from building_gan_amd import models as hm
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.train import Trainer
    from oracle import networks as om
    from oracle.objectives import train_iteration
    cpu, gpu, sz, Go, _, _, _, noise, pz, vz = _setup("small", 23, True, cuda)
    act = "sigmoid" if gan_loss == "NSGAN" else ""
    torch.manual_seed(23)
    Do = om.DiscriminatorVoxel(7, 6, 128, 12, act)
    tr = Trainer(cuda, gan_loss=gan_loss)
    tr.generator.load_state_dict(Go.state_dict())
    tr.discriminator.load_state_dict(Do.state_dict())
    og = torch.optim.Adam(Go.parameters(), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(Do.parameters(), lr=1e-4, betas=(0.5, 0.999))
    before = {n: p.detach().clone() for n, p in list(Do.named_parameters()) + list(Go.named_parameters())}
    so = train_iteration(Go, Do, og, od, cpu, detach_batch, 0, pz, vz, noise, None, gan_loss=gan_loss)
    sh = tr.train_iteration(gpu, 0, pz.to(cuda), vz.to(cuda), _to(noise, cuda), None)
    assert "gp" not in sh
    torch.testing.assert_close(sh["d_loss"].cpu(), so["d_loss"], rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(sh["g_loss"].cpu().view(-1), so["g_loss"].view(-1), rtol=1e-2, atol=5e-3)
    _assert_adam_steps_agree(Do.named_parameters(), tr.discriminator.named_parameters(), before)
    _assert_adam_steps_agree(Go.named_parameters(), tr.generator.named_parameters(), before)


@pytest.mark.parametrize("preset,ragged,seed", [("small", True, 31), ("small", "jitter", 32), ("dataset", "jitter", 33)])
def test_critic_head_in_one_host_call_equals_op_by_op_path(cuda, preset, ragged, seed, d_basis):
    """ops.CriticHead (bg_critic_head_forward_f32 / _backward_f32: everything of a WGAN-GP critic update below the
    coefficient table, forward and hand-derived backward, as one host call each) against critic_terms + CriticLoss
    (the same kernels launched op by op under autograd) on identical parameters, labels and sample point: objective
    and penalty to 1e-6 relative, every discriminator gradient to 3e-5 of its scale (two fp32 evaluations that add in
    different orders: 2.0e-5 was measured on a bias gradient, a sum with cancellations)."""
    from building_gan_amd import models as hm, ops
    from building_gan_amd.plan import build_plan
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, seed, ragged, cuda)
    torch.manual_seed(seed)
    Nv = sz["Nv"]
    fake = (torch.zeros(Nv, 6).scatter_(1, torch.randint(0, 6, (Nv, 1)), 1.0) * (torch.rand(Nv, 1) > 0.3)).to(cuda)
    sample = ops.gp_sample_raw(gpu.voxel_label, fake, torch.rand(Nv, device=cuda))
    plan = build_plan(gpu)
    B, S = plan.B, plan.S
    wb = torch.cat([torch.full((B,), -1.0 / B), torch.full((B,), 1.0 / B)]).to(cuda)
    ws = torch.cat([torch.full((S,), -1.0 / S), torch.full((S,), 1.0 / S)]).to(cuda)
    coef = 10.0 / Nv

    def grads_of(fn):
        for p in Dh.parameters():
            p.grad = None
        plan.begin_iteration()
        with Dh.shared_coefficients():
            loss, pen = fn()
        loss.backward()
        return loss.detach().cpu(), pen.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in Dh.named_parameters()}

    def op_by_op():
        vb, vs, g = Dh.critic_terms(gpu, gpu.voxel_label, fake, sample, stacked=True)
        return ops.CriticLoss.apply(vb, vs, g, wb, ws, coef)

    def one_call():
        res = Dh.critic_objective(gpu, gpu.voxel_label, fake, sample, wb, ws, coef)
        assert res is not None, "the one-call path declined this batch"
        return res

    l0, p0, g0 = grads_of(op_by_op)
    l1, p1, g1 = grads_of(one_call)
    torch.testing.assert_close(l1, l0, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(p1, p0, rtol=1e-6, atol=1e-7)
    top = max(float(g.abs().max()) for g in g0.values())
    worst = 0.0
    for n in g0:
        e = float((g1[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-3 * top)
        assert e <= 3e-5, (n, e)
        worst = max(worst, e)
    print("critic head vs op-by-op: worst gradient error %.2e" % worst)
    # switched off, the trainer's iteration takes the op-by-op path and arrives at the same update
    assert hm.FUSE_CRITIC_HEAD


def test_trainer_iteration_with_and_without_the_one_call_critic_head(cuda, monkeypatch):
    """One critic-only Trainer iteration from identical state with models.FUSE_CRITIC_HEAD on / off: same d_loss, same
    flat gradient buffer (to 2e-5 of its max), same penalty statistic."""
    from building_gan_amd import models as hm
    from building_gan_amd.train import Trainer
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 35, "jitter", cuda)
    u = torch.rand(sz["Nv"], device=cuda)
    res = []
    for on in (True, False):
        monkeypatch.setattr(hm, "FUSE_CRITIC_HEAD", on)
        torch.manual_seed(35)
        tr = Trainer(cuda)
        tr.generator.load_state_dict(Go.state_dict())
        tr.discriminator.load_state_dict(Do.state_dict())
        tr.refresh_derived()
        gpu.__dict__.pop("_bg_plan", None)
        st = tr.train_iteration(gpu, 1, pz.to(cuda), vz.to(cuda), _to(noise, cuda), u)
        res.append((st["d_loss"].cpu(), st["gp"].cpu(), tr.opt_d.grad.detach().cpu().clone()))
    (l1, gp1, g1), (l0, gp0, g0) = res
    torch.testing.assert_close(l1, l0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gp1, gp0, rtol=1e-5, atol=1e-7)
    assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())


@pytest.mark.parametrize("preset,ragged,seed", [("small", "jitter", 61), ("dataset", "jitter", 62)])
def test_critic_update_without_an_autograd_pass_equals_the_autograd_path(cuda, preset, ragged, seed):
    """Trainer.manual_critic (the fused critic update's three nodes called forward and backward directly, gradients
    assigned to .grad) against the same nodes under autograd, over three iterations incl. one with the generator
    update: the same launches in the same order, so losses, flat gradient buffers and parameters are bitwise equal."""
    from building_gan_amd.train import Trainer
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, seed, ragged, cuda)
    u = torch.rand(sz["Nv"], generator=torch.Generator().manual_seed(seed)).to(cuda)
    res = []
    for manual in (True, False):
        torch.manual_seed(seed)
        tr = Trainer(cuda)
        tr.manual_critic = manual
        tr.generator.load_state_dict(Go.state_dict())
        tr.discriminator.load_state_dict(Do.state_dict())
        tr.refresh_derived()
        gpu.__dict__.pop("_bg_plan", None)
        out = []
        for i in (1, 2, 5):
            st = tr.train_iteration(gpu, i, pz.to(cuda), vz.to(cuda), _to(noise, cuda), u)
            out.append((st["d_loss"].cpu(), st["gp"].cpu(), tr.opt_d.grad.detach().cpu().clone()))
        res.append((out, tr.opt_d.flat.detach().cpu().clone(), tr.opt_g.flat.detach().cpu().clone()))
    (o1, d1, g1), (o0, d0, g0) = res
    for (l1, p1, q1), (l0, p0, q0) in zip(o1, o0):
        assert torch.equal(l1, l0) and torch.equal(p1, p0) and torch.equal(q1, q0)
    assert torch.equal(d1, d0) and torch.equal(g1, g0)
    assert bool(torch.isfinite(d1).all())


@pytest.mark.parametrize("preset,ragged,seed", [("small", True, 51), ("small", "jitter", 53), ("dataset", "jitter", 52)])
def test_generator_stack_in_one_host_call_equals_node_by_node_path(cuda, preset, ragged, seed, monkeypatch):
    """ops.VoxelStackG (bg_gstack_forward_f32 / _backward_f32: the generator's voxel GNN as one autograd node, one host
    call each way) against the node-per-call path (ops.GBlock / ops.AttentionNode) on identical parameters, inputs and
    noise: the four outputs bit-identical (same launches), every generator gradient to 2e-5 of its scale (weight
    gradients are summed over the calls in the same launches; the program features' gradient in another order)."""
    from building_gan_amd import models as hm
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, seed, ragged, cuda)
    torch.manual_seed(seed)
    Nv, Ec = sz["Nv"], sz["Ec"]
    w_out, w_soft = torch.rand(Nv, 6, device=cuda) - 0.5, torch.rand(Nv, 6, device=cuda) - 0.5
    w_ms, w_mh = torch.rand(Nv, 1, device=cuda) - 0.5, torch.rand(Nv, 1, device=cuda) - 0.5
    w_as, w_ah = torch.rand(Ec, 1, device=cuda) - 0.5, torch.rand(Ec, 1, device=cuda) - 0.5
    res = []
    for on in (True, False):
        monkeypatch.setattr(hm, "FUSE_GSTACK", on)
        for p in Gh.parameters():
            p.grad = None
        out, soft_out, mask, att, idx = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))
        loss = ((out * w_out).sum() + (soft_out * w_soft).sum() + (mask["soft"] * w_ms).sum() + (mask["hard"] * w_mh).sum()
                + (att["soft"] * w_as).sum() + (att["hard"] * w_ah).sum())
        loss.backward()
        res.append(([t.detach().clone() for t in (out, soft_out, mask["soft"], mask["hard"], att["soft"], att["hard"], idx)],
                    {n: p.grad.detach().clone() for n, p in Gh.named_parameters() if p.grad is not None}))
    (o1, g1), (o0, g0) = res
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    assert set(g1) == set(g0)
    top = max(float(g.abs().max()) for g in g0.values())
    worst = 0.0
    for n in g0:
        e = float((g1[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-3 * top)
        assert e <= 2e-5, (n, e)
        worst = max(worst, e)
    print("generator stack vs node by node: worst gradient error %.2e" % worst)


def test_graph_capture_after_eager_iterations_on_other_batches(cuda):
    """bench.py's order: eager iterations over a stream of differently shaped batches, then Trainer.capture on a fixed
    one.  An autograd node that kept its own outputs alive in a reference cycle (and with them the parameters'
    AccumulateGrad nodes of the eager iterations, made on the default stream) pulled the default stream into the
    capture and crashed hipStreamEndCapture; nodes keep outputs through save_for_backward, and capture() collects
    dead cycles first."""
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import make_batch
    from building_gan_amd.train import Trainer
    torch.manual_seed(3)
    tr = Trainer(cuda)
    others = [make_batch("small", seed=60 + k, ragged="jitter").to(cuda) for k in range(3)]
    for i in range(6):
        b = others[i % 3]
        b.__dict__.pop("_bg_plan", None)
        build_plan(b)
        tr.train_iteration(b, i)
    fixed = make_batch("small", seed=7).to(cuda)
    build_plan(fixed)
    tr.capture(fixed)
    for i in range(6):
        st = tr.train_iteration_graphed(i)
    torch.cuda.synchronize()
    assert torch.isfinite(tr.opt_d.grad).all() and torch.isfinite(tr.opt_g.grad).all()


@pytest.mark.parametrize("preset,ragged,seed", [("small", True, 61), ("dataset", "jitter", 62)])
def test_program_gnn_in_one_host_call_equals_op_by_op_path(cuda, preset, ragged, seed, monkeypatch):
    """ops.ProgramStack (bg_pstack_forward_f32 / _backward_f32) against the op-by-op autograd ProgramGNN on identical
    parameters and inputs: output rtol 2e-6 of its scale (the class-slot term is scaled once as ratio / count instead of
    mean x ratio), every gradient to 2e-5 of its scale."""
    from building_gan_amd import models as hm
    from building_gan_amd.plan import build_plan
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, seed, ragged, cuda)
    plan = build_plan(gpu)
    torch.manual_seed(seed)
    wgt = torch.rand(sz["Np"], 128, device=cuda) - 0.5
    pg = Gh.programGNN
    res = []
    for on in (True, False):
        monkeypatch.setattr(hm, "FUSE_PSTACK", on)
        for p in pg.parameters():
            p.grad = None
        x = pg(gpu.program_class_feature, gpu.story_level_feature, gpu.program_target_ratio, pz.to(cuda), plan)
        assert (type(x.grad_fn).__name__ == "ProgramStackBackward") == on
        (x * wgt).sum().backward()
        res.append((x.detach().clone(), {n: p.grad.detach().clone() for n, p in pg.named_parameters()}))
    (x1, g1), (x0, g0) = res
    assert float((x1 - x0).abs().max()) <= 2e-6 * float(x0.abs().max())
    top = max(float(g.abs().max()) for g in g0.values())
    for n in g0:
        e = float((g1[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-3 * top)
        assert e <= 2e-5, (n, e)


@pytest.mark.parametrize("preset,ragged,seed", [("small", True, 71), ("dataset", "jitter", 72)])
def test_generator_objective_in_one_host_call_equals_op_by_op_path(cuda, preset, ragged, seed, d_basis):
    """ops.GLossHead (bg_gloss_head_forward_f32 / _backward_f32) against discriminator(batch, label) + the sums of
    VolumetricDesignLoss_G under autograd, the discriminator frozen: value rtol 1e-6, d / d label to 2e-5 of its max."""
    from building_gan_amd.plan import build_plan
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, seed, ragged, cuda)
    torch.manual_seed(seed)
    Nv = sz["Nv"]
    plan = build_plan(gpu)
    B, S = plan.B, plan.S
    for p in Dh.parameters():
        p.requires_grad = False
    wb = torch.full((2 * B,), -1.0 / B, device=cuda)
    ws = torch.full((2 * S,), -1.0 / S, device=cuda)
    base = torch.rand(Nv, 6, device=cuda)
    la = base.clone().requires_grad_(True)
    vb, vs = Dh(gpu, la)
    l0 = -vb.sum() / B - vs.sum() / S
    l0.backward()
    plan.begin_iteration()
    lb = base.clone().requires_grad_(True)
    l1 = Dh.generator_objective(gpu, lb, wb, ws)
    assert l1 is not None, "the one-call path declined this batch"
    l1.backward()
    torch.testing.assert_close(l1.detach().cpu(), l0.detach().cpu(), rtol=1e-6, atol=1e-7)
    assert float((lb.grad - la.grad).abs().max()) <= 2e-5 * float(la.grad.abs().max())
    assert float(la.grad.abs().max()) > 0


def test_fused_attention_forward_equals_unfused(cuda):
    """The one-launch Attention call of graph-free passes (bg_attention_fwd_f32) against the five-kernel
    autograd path on identical weights, inputs and noise: same pointers, activations to rounding."""
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 41, True, cuda)
    from building_gan_amd import models as hm
    nz = _to(noise, cuda)
    hm.FUSE_ATTENTION_NODE = hm.FUSE_GBLOCK = False
    try:
        ref = Gh(gpu, pz.to(cuda), vz.to(cuda), nz)                # builds a graph, op by op -> unfused kernels
    finally:
        hm.FUSE_ATTENTION_NODE = hm.FUSE_GBLOCK = True
    with torch.no_grad():
        got = Gh(gpu, pz.to(cuda), vz.to(cuda), nz)                # fused kernel
    out_r, soft_r, mask_r, att_r, idx_r = ref
    out_g, soft_g, mask_g, att_g, idx_g = got
    assert torch.equal(idx_g, idx_r)
    assert torch.equal(att_g["hard"].round(), att_r["hard"].round()) and torch.equal(mask_g["hard"].round(), mask_r["hard"].round())
    for a, b in ((out_g, out_r), (soft_g, soft_r), (mask_g["soft"], mask_r["soft"]), (att_g["soft"], att_r["soft"])):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("preset,ragged", [("small", True), ("dataset", "jitter")])
def test_generator_forward_in_one_host_call_equals_op_by_op_path(cuda, preset, ragged):
    """bg_generator_forward_f32 (the whole graph-free forward issued by one C call) against the op-by-op host path:
    the same kernels with the same arguments, so every output must agree bit for bit -- with explicit noise, with
    noise drawn inside (same generator seed), and with the kept derived weights the trainer passes in."""
    from building_gan_amd import models as hm
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup(preset, 57, ragged, cuda)
    nz, pzd = _to(noise, cuda), pz.to(cuda)
    with torch.no_grad():
        derived = [t.clone() for t in Gh.voxelGNN.derived_weights()]
    res = {}
    for fused in (True, False):
        hm.FUSE_GENERATOR_FORWARD = fused
        try:
            with torch.no_grad():
                a = Gh(gpu, pzd, None, nz)
                torch.manual_seed(5)
                b = Gh(gpu, pzd, None, None)
                c = Gh(gpu, pzd, None, nz, derived)
            res[fused] = (a, b, c)
        finally:
            hm.FUSE_GENERATOR_FORWARD = True
    for ra, rb in zip(res[True], res[False]):
        out_a, soft_a, mask_a, att_a, idx_a = ra
        out_b, soft_b, mask_b, att_b, idx_b = rb
        assert torch.equal(idx_a, idx_b) and idx_a.dtype == torch.int64
        assert torch.equal(out_a, out_b) and torch.equal(soft_a, soft_b)
        for k in ("hard", "soft"):
            assert mask_a[k].shape == mask_b[k].shape and att_a[k].shape == att_b[k].shape
            assert torch.equal(mask_a[k], mask_b[k]) and torch.equal(att_a[k], att_b[k])
    assert not torch.equal(res[True][0][3]["soft"], res[True][1][3]["soft"])          # other noise, other pointer weights


def test_generator_block_node_equals_op_by_op_path(cuda):
    """ops.GBlock (one autograd node, hand-written backward) against the op-by-op composed block on identical
    weights and inputs: output and every gradient (input, messageMLP, updateMLP) to fp32 rounding."""
    from building_gan_amd import models as hm
    from building_gan_amd.plan import build_plan
    cpu, gpu, sz, *_ = _setup("small", 51, True, cuda)
    plan = build_plan(gpu)
    assert plan.vox_all_has
    torch.manual_seed(51)
    blk = hm.VoxelBlock(128, if_agg_story_in_update=True).to(cuda)
    pdm4 = hm.position_terms(gpu)
    v0 = (torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1)
    gout = torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1
    from building_gan_amd import ops
    res = {}
    for fused in (True, "posb", False):
        hm.FUSE_GBLOCK = bool(fused)
        try:
            for p in blk.parameters():
                p.grad = None
            v = v0.clone().requires_grad_(True)
            comp = blk.compose()
            # "posb": the position + bias term evaluated once, the product starting from it (K = 2H)
            posb = ops.block_pos_term(pdm4, comp[0], comp[1]) if fused == "posb" else None
            out = blk.forward_composed(v, plan, pdm4, comp, posb=posb)
            (out * gout).sum().backward()
            res[fused] = [out.detach(), v.grad] + [p.grad.clone() for p in blk.parameters()]
        finally:
            hm.FUSE_GBLOCK = True
    for key in (True, "posb"):
        for a, b in zip(res[key], res[False]):
            scale = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= 2e-5 * scale
    with torch.no_grad():       # and the graph-free pass with / without the precomputed term
        comp = blk.compose()
        a = blk.forward_composed(v0, plan, pdm4, comp)
        b = blk.forward_composed(v0, plan, pdm4, comp, posb=ops.block_pos_term(pdm4, comp[0], comp[1]))
    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())


def test_attention_node_equals_op_by_op_path(cuda):
    """ops.AttentionNode (fused forward, hand-written backward) against the op-by-op Attention call on identical
    weights, inputs and noise: all five outputs and every gradient (voxel features, program features, NN_v,
    NN_p, decoder, theta), for a loss that uses every output."""
    from building_gan_amd import models as hm
    from building_gan_amd.plan import build_plan
    cpu, gpu, sz, *_ = _setup("small", 52, True, cuda)
    plan = build_plan(gpu)
    torch.manual_seed(52)
    att = hm.Attention(128).to(cuda)
    v0 = torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1
    x0 = torch.rand(sz["Np"], 128, device=cuda) * 2 - 1
    noise = (torch.empty(sz["Nv"], 2, device=cuda).exponential_(), torch.empty(sz["Ec"], device=cuda).exponential_())
    w = [torch.rand(sz["Nv"], 1, device=cuda), torch.rand(sz["Nv"], 1, device=cuda), torch.rand(sz["Ec"], 1, device=cuda),
         torch.rand(sz["Ec"], 1, device=cuda), torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1]
    res = {}
    for fused in (True, False):
        hm.FUSE_ATTENTION_NODE = fused
        try:
            for p in att.parameters():
                p.grad = None
            v, x = v0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
            mask, a, nv = att(x, v, plan, noise)
            outs = [mask["soft"], mask["hard"], a["soft"], a["hard"], nv]
            sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
            res[fused] = [o.detach() for o in outs] + [v.grad, x.grad] + [p.grad.clone() for p in att.parameters()]
        finally:
            hm.FUSE_ATTENTION_NODE = True
    assert torch.equal(res[True][1].round(), res[False][1].round()) and torch.equal(res[True][3].round(), res[False][3].round())
    for a_, b_ in zip(res[True], res[False]):
        scale = float(b_.abs().max()) + 1e-12
        assert float((a_ - b_).abs().max()) <= 5e-5 * scale


def test_shared_weight_gradient_buffers_equal_autograd_sums(cuda):
    """models.SHARE_WEIGHT_GRADS (the 12 block calls and 6 attention calls add their weight gradients to one
    buffer inside the producing launches) against autograd's own per-call gradients summed by AccumulateGrad:
    every generator parameter gradient to fp32 rounding (the order of the 12 / 6 additions differs), twice in a
    row on the same module (the buffers are per forward)."""
    from building_gan_amd import models as hm
    from building_gan_amd.losses import VolumetricDesignLoss_G
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 53, True, cuda)
    res = {}
    for share in (True, False, True):
        hm.SHARE_WEIGHT_GRADS = share
        try:
            for p in Gh.parameters():
                p.grad = None
            out_h, soft_h, mask_h, att_h, _ = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))
            VolumetricDesignLoss_G()(Dh(gpu, out_h), gpu, att_h["hard"], mask_h["hard"], 6)[0].backward()
            got = [p.grad.clone() for p in Gh.parameters()]
        finally:
            hm.SHARE_WEIGHT_GRADS = True
        if share in res:
            for a, b in zip(got, res[share]):
                assert torch.equal(a, b)            # deterministic, and nothing left over from the first pass
        res[share] = got
    for (n, _), a, b in zip(Gh.named_parameters(), res[True], res[False]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale, n


def test_shared_weight_gradient_buffers_fail_loudly_on_partial_backward(cuda):
    """the first call's node hands the shared buffers to autograd.  A second backward over a retained
    graph and torch.autograd.grad restricted to one parameter must give the same numbers as the plain backward; a
    backward that stops short of the first call (inputs = an intermediate activation) or a chain cut by a detach
    must raise instead of dropping the accumulated weight gradients."""
    from building_gan_amd import models as hm, ops
    from building_gan_amd._lib import BganError
    from building_gan_amd.plan import build_plan
    cpu, gpu, sz, *_ = _setup("small", 56, True, cuda)
    plan = build_plan(gpu)
    torch.manual_seed(56)
    blk = hm.VoxelBlock(128, if_agg_story_in_update=True).to(cuda)
    pdm4 = hm.position_terms(gpu)
    v0 = torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1
    w = torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1

    def chain(cut=False):
        sg = ops.SharedGrads()
        comp = blk.compose()
        v = v0.clone().requires_grad_(True)
        mids = [v]
        for i in range(3):
            v = blk.forward_composed(v, plan, pdm4, comp, (sg, i == 0))
            if cut and i == 0:
                v = v.detach().requires_grad_(True)
            mids.append(v)
        return (v * w).sum(), mids

    loss, _ = chain()
    for p in blk.parameters():
        p.grad = None
    loss.backward(retain_graph=True)
    first = [p.grad.clone() for p in blk.parameters()]
    for p in blk.parameters():
        p.grad = None
    loss.backward()                                            # second pass over the retained graph
    for a, b in zip(first, [p.grad for p in blk.parameters()]):
        assert torch.equal(a, b)
    loss, _ = chain()
    Wm = blk.messageMLP[0].weight
    (only,) = torch.autograd.grad(loss, [Wm])                  # restricted to one parameter
    assert torch.equal(only, first[[id(p) for p in blk.parameters()].index(id(Wm))])
    loss, mids = chain()
    with pytest.raises(BganError, match="shared weight gradients"):
        torch.autograd.grad(loss, [mids[2]])                   # stops before the first call's node
    loss, _ = chain(cut=True)
    with pytest.raises(BganError, match="shared weight gradients"):
        loss.backward()                                        # the first call is cut off from the loss


def test_coefficient_inputs_by_one_launch_equal_cat_construction(cuda, d_basis, monkeypatch):
    """ops.PolyCoefD (X0 / E written into the recurrence's buffers by bg_d_coef_init_f32, parameter gradients
    read out by bg_d_coef_init_bwd_f32) against the cat / transpose construction feeding ops.PolyCoef: the
    collapsed weight bit for bit, every discriminator gradient to fp32 rounding (the sum over the layers'
    slabs runs in a different order)."""
    from building_gan_amd import models as hm
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 54, True, cuda)
    monkeypatch.setattr(hm, "FUSE_COMPOSE_D", False)
    res = {}
    for fused in (True, False):
        hm.FUSE_COEF_INPUTS = fused
        try:
            for p in Dh.parameters():
                p.grad = None
            W = Dh.voxelGNN_D.coefficients(gpu.voxel_feature.shape[1] - 3, 32, 8)
            lab = gpu.voxel_label.clone().requires_grad_(True)
            b, s = Dh(gpu, lab)
            (b.sum() + 2 * s.sum()).backward()
            res[fused] = [W.detach().clone(), b.detach(), s.detach(), lab.grad] + \
                         [p.grad.clone() for p in Dh.parameters()]
        finally:
            hm.FUSE_COEF_INPUTS = True
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
    names = ["W", "b", "s", "dlabel"] + [n for n, _ in Dh.named_parameters()]
    for n, a, r in zip(names, res[True], res[False]):
        scale = float(r.abs().max()) + 1e-12
        assert float((a - r).abs().max()) <= 2e-5 * scale, n


def test_coefficient_recurrence_in_one_host_call_equals_separate_launches(cuda, d_basis, monkeypatch):
    """ops.FUSE_COEF_RECURRENCE (bg_d_coef_forward_f32 / bg_d_coef_backward_f32: the recurrence's launches issued by
    one C call each way) against the same launches issued one by one from the host: the collapsed weight, the
    critic's outputs and every discriminator gradient bit for bit."""
    from building_gan_amd import models as hm, ops
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 58, True, cuda)
    monkeypatch.setattr(hm, "FUSE_COMPOSE_D", False)      # (its own test: test_block_composition_inside_the_coefficient_node)
    res = {}
    for fused in (True, False):
        ops.FUSE_COEF_RECURRENCE = fused
        try:
            for p in Dh.parameters():
                p.grad = None
            W = Dh.voxelGNN_D.coefficients(gpu.voxel_feature.shape[1] - 3, 32, 8)
            lab = gpu.voxel_label.clone().requires_grad_(True)
            b, s = Dh(gpu, lab)
            (b.sum() + 2 * s.sum()).backward()
            res[fused] = [W.detach().clone(), b.detach(), s.detach(), lab.grad] + [p.grad.clone() for p in Dh.parameters()]
        finally:
            ops.FUSE_COEF_RECURRENCE = True
    names = ["W", "b", "s", "dlabel"] + [n for n, _ in Dh.named_parameters()]
    for n, a, r in zip(names, res[True], res[False]):
        assert torch.equal(a, r), n


def test_block_composition_inside_the_coefficient_node(cuda, d_basis):
    """models.FUSE_COMPOSE_D (ops.PolyCoefDW: compose() and the recurrence as one node, bg_d_table_forward_f32 /
    _backward_f32) against compose() under autograd feeding ops.PolyCoefD: the table and the critic's outputs bit for bit,
    every discriminator gradient to fp32 rounding."""
    from building_gan_amd import models as hm
    cpu, gpu, sz, Go, Do, Gh, Dh, noise, pz, vz = _setup("small", 59, True, cuda)
    res = {}
    for fused in (True, False):
        hm.FUSE_COMPOSE_D = fused
        try:
            for p in Dh.parameters():
                p.grad = None
            W = Dh.voxelGNN_D.coefficients(gpu.voxel_feature.shape[1] - 3, 32, 8)
            lab = gpu.voxel_label.clone().requires_grad_(True)
            b, s = Dh(gpu, lab)
            (b.sum() + 2 * s.sum()).backward()
            res[fused] = [W.detach().clone(), b.detach(), s.detach(), lab.grad] + [p.grad.clone() for p in Dh.parameters()]
        finally:
            hm.FUSE_COMPOSE_D = True
    names = ["W", "b", "s", "dlabel"] + [n for n, _ in Dh.named_parameters()]
    for n, a, r in zip(names[:4], res[True], res[False]):
        assert torch.equal(a, r), n
    worst = 0.0
    for n, a, r in zip(names[4:], res[True][4:], res[False][4:]):
        e = float((a - r).abs().max()) / (float(r.abs().max()) + 1e-12)
        worst = max(worst, e)
        assert e <= 2e-6, (n, e)
    print("composition inside the node vs compose() under autograd: worst gradient difference %.1e" % worst)


def test_attention_backward_head_one_launch_equals_separate_launches(cuda):
    """bg_attention_bwd_head_f32 (the head of ops.AttentionNode's backward in one launch) against the same node
    with the head as its seven separate launches: every gradient to fp32 rounding, with all five outputs used
    and with only the voxel features used (absent edge / mask gradients)."""
    from building_gan_amd import models as hm, ops
    from building_gan_amd.plan import build_plan
    cpu, gpu, sz, *_ = _setup("small", 55, True, cuda)
    plan = build_plan(gpu)
    torch.manual_seed(55)
    att = hm.Attention(128).to(cuda)
    v0 = torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1
    x0 = torch.rand(sz["Np"], 128, device=cuda) * 2 - 1
    noise = (torch.empty(sz["Nv"], 2, device=cuda).exponential_(), torch.empty(sz["Ec"], device=cuda).exponential_())
    w = [torch.rand(sz["Nv"], 1, device=cuda), torch.rand(sz["Nv"], 1, device=cuda), torch.rand(sz["Ec"], 1, device=cuda),
         torch.rand(sz["Ec"], 1, device=cuda), torch.rand(sz["Nv"], 128, device=cuda) * 2 - 1]
    for used in ((0, 1, 2, 3, 4), (4,)):
        res = {}
        for fused in (True, False):
            ops.FUSE_ATTENTION_BWD_HEAD = fused
            try:
                for p in att.parameters():
                    p.grad = None
                v, x = v0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
                mask, a, nv = att(x, v, plan, noise)
                outs = [mask["soft"], mask["hard"], a["soft"], a["hard"], nv]
                sum((outs[i] * w[i]).sum() for i in used).backward()
                res[fused] = [v.grad, x.grad] + [p.grad.clone() for p in att.parameters() if p.grad is not None]
            finally:
                ops.FUSE_ATTENTION_BWD_HEAD = True
        assert len(res[True]) == len(res[False])
        for a_, b_ in zip(res[True], res[False]):
            scale = float(b_.abs().max()) + 1e-12
            assert float((a_ - b_).abs().max()) <= 5e-6 * scale
