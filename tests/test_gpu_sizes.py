"""Parity at the sizes BASELINE.json names ("configs_untested"):

  configs[2]  10^6-edge message passing: gather-mean BIT-EXACT against the oracle's scatter-mean, H = 128 and
              H = 256 (the latter takes the non-temporal-store path of the kernel: result >= 128 MB);
  configs[4]  one 32-story 64 x 64 building (131,072 voxels, max_len = 32): generator forward against the oracle;
              8 copies (1,048,576 voxels): the structural properties of batched variation sampling;
  configs[1]  at the `dataset` preset: generator forward, the flat gradient buffers of one Trainer iteration
              (critic phase and generator phase) against the oracle's .grad tensors, and the float64 treatment
              of the generator backward and of the gradient penalty's double backward.
Tolerances sit next to the measured errors (printed by the tests; `pytest -s` shows them).
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    """max |a - b| / max |b|  (both on the CPU, b the reference)."""
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


# ------------------------------------------------------------------------------------ configs[2]
@pytest.mark.parametrize("H", [128, 256])
def test_gather_mean_bit_exact_at_million_edges(cuda, H):
    from building_gan_amd import ops
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from oracle.scatter import scatter_mean
    cpu = make_batch("microbench", seed=0)
    sz = batch_sizes(cpu)
    assert sz["Ev"] == 1006896 and sz["Nv"] == 174240
    gpu = copy.deepcopy(cpu).to(cuda)
    plan = build_plan(gpu)
    torch.manual_seed(H)
    x = torch.rand(sz["Nv"], H) * 2 - 1
    ref = scatter_mean(x.index_select(0, cpu.voxel_edge[0]), cpu.voxel_edge[1], sz["Nv"])
    out = ops.spmm_raw(plan.vox_mean, x.to(cuda))
    if H == 256:
        assert out.numel() * 4 >= 64 << 20             # the non-temporal-store branch of the gather kernels
    assert torch.equal(out.cpu(), ref)
    # the transposed operator (the backward of the same layer) against the dense identity <A x, y> = <x, A^T y>
    y = torch.rand(sz["Nv"], H) * 2 - 1
    outT = ops.spmm_raw(plan.vox_mean.T, y.to(cuda)).cpu()
    lhs, rhs = float((ref.double() * y.double()).sum()), float((x.double() * outT.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0) + 1e-3


def test_plans_bit_exact_at_million_edges(cuda):
    import numpy as np
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import make_batch
    from oracle import plan as oplan
    cpu = make_batch("microbench", seed=1)
    gpu = copy.deepcopy(cpu).to(cuda)
    p = build_plan(gpu)
    ve = cpu.voxel_edge.numpy()
    rp, col, perm = oplan.csr_by_key(ve[1], ve[0], p.Nv)
    c = p.vox_mean.csr
    assert np.array_equal(c.rowptr.cpu().numpy(), rp) and np.array_equal(c.col.cpu().numpy(), col)
    assert np.array_equal(c.perm.cpu().numpy(), perm)
    rp, col, perm = oplan.csr_by_key(ve[0], ve[1], p.Nv)
    c = p.vox_mean.T.csr
    assert np.array_equal(c.rowptr.cpu().numpy(), rp) and np.array_equal(c.col.cpu().numpy(), col)


# ------------------------------------------------------------------------------------ configs[4]
def _nets(dev, max_len=20, seed=0):
    from building_gan_amd import models as hm
    from oracle import networks as om
    torch.manual_seed(seed)
    Go, Do = om.Generator(7, 7, 128, 32, 4, 12, max_len=max_len), om.DiscriminatorVoxel(7, 6, 128, 12, max_len=max_len)
    Gh = hm.Generator(7, 7, 128, 32, 4, 12, dev, max_len=max_len).to(dev)
    Dh = hm.DiscriminatorVoxel(7, 6, 128, 12, max_len=max_len).to(dev)
    Gh.load_state_dict(Go.state_dict())
    Dh.load_state_dict(Do.state_dict())
    return Go, Do, Gh, Dh


def _noise(sz, n=6):
    return [(torch.empty(sz["Nv"], 2).exponential_(), torch.empty(sz["Ec"]).exponential_()) for _ in range(n)]


def _to(noise, dev):
    return [(a.to(dev), b.to(dev)) for a, b in noise]


def _compare_generator_outputs(rh, ro, tag, soft_tol=1e-4, flip_budget=2e-4):
    """Continuous outputs to rtol 1e-4 of their scale (north_star); the discrete pointer agrees wherever the two
    best cross edges of a voxel are not within rounding of each other (measured flip fraction printed)."""
    out_o, soft_o, mask_o, att_o, idx_o = ro
    out_h, soft_h, mask_h, att_h, idx_h = rh
    e = {"mask_soft": _rel(mask_h["soft"].cpu(), mask_o["soft"]), "att_soft": _rel(att_h["soft"].cpu(), att_o["soft"])}
    flips = float((idx_h.cpu() != idx_o).float().mean())
    mflips = float((mask_h["hard"].cpu().round() != mask_o["hard"].round()).float().mean())
    same = (idx_h.cpu() == idx_o) & (mask_h["hard"].cpu().round() == mask_o["hard"].round()).view(-1)
    e["soft_out_where_same_pointer"] = _rel(soft_h.cpu()[same], soft_o[same])
    print("%s: relative errors %s; pointer flips %.2e, mask flips %.2e of %d voxels" % (tag, e, flips, mflips, idx_o.numel()))
    assert e["mask_soft"] <= soft_tol and e["att_soft"] <= soft_tol and e["soft_out_where_same_pointer"] <= soft_tol
    assert flips <= flip_budget and mflips <= flip_budget
    return flips


def test_large_building_generator_forward_parity(cuda):
    """One 32 x 64 x 64 building (BASELINE configs[4]; >65,536 voxels: separate gather-mean / story launches,
    128 x 64 GEMM tiles) against the oracle on the same weights and noise."""
    from building_gan_amd.synthetic import batch_sizes, make_batch
    cpu = make_batch("large", seed=0)
    sz = batch_sizes(cpu)
    assert sz["Nv"] == 131072 and sz["S"] == 32
    Go, _, Gh, _ = _nets(cuda, max_len=32)
    torch.manual_seed(5)
    noise, pz, vz = _noise(sz), torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32)
    gpu = copy.deepcopy(cpu).to(cuda)
    with torch.no_grad():
        ro = Go(cpu, pz, vz, noise)
        rh = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))
    _compare_generator_outputs(rh, ro, "large building")


def test_million_voxel_variation_batch_properties(cuda):
    """8 variations of the 32-story building in one batch (1,048,576 voxels): every copy points into its own
    program nodes and its own story, labels are one-hot or void, ratios are normalised, and a copy that gets
    the SAME noise as copy 0 reproduces copy 0's design bit for bit (graphs never interact)."""
    from building_gan_amd import models as hm
    from building_gan_amd.graph import Batch
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.util_graph import get_program_ratio
    one = make_batch("large", seed=0)
    n1, np1, ec1 = one.voxel_feature.shape[0], one.program_class_feature.shape[0], one.cross_edge_voxel_index_select.shape[0]
    V = 8
    batch = Batch.from_data_list([one] * V).to(cuda)
    sz = batch_sizes(batch)
    assert sz["Nv"] == 1048576
    torch.manual_seed(2)
    G = hm.Generator(7, 7, 128, 32, 4, 12, cuda, max_len=32).to(cuda)
    pz = torch.rand(sz["Np"], 32, device=cuda)
    pz[np1:2 * np1] = pz[:np1]
    noise = []
    for _ in range(6):
        em, ea = torch.empty(sz["Nv"], 2, device=cuda).exponential_(), torch.empty(sz["Ec"], device=cuda).exponential_()
        em[n1:2 * n1], ea[ec1:2 * ec1] = em[:n1], ea[:ec1]
        noise.append((em, ea))
    with torch.no_grad():
        out, soft_out, mask, att, idx = G(batch, pz, None, noise)
        ratio, _, far = get_program_ratio(batch, att["hard"], mask["hard"], 6)
    assert torch.isfinite(soft_out).all() and out.shape == (sz["Nv"], 6)
    copy_of = torch.arange(sz["Nv"], device=cuda) // n1
    assert torch.equal(idx // np1, copy_of)                                    # own program nodes
    story_of_program = batch.program_floor_cluster
    assert torch.equal(story_of_program[idx], batch.voxel_floor_cluster)       # of its own story
    rows = out.round().sum(1)
    assert bool(((rows == 0) | (rows == 1)).all())
    assert torch.equal(rows == 1, mask["hard"].view(-1).round() == 1)
    assert torch.equal(idx[n1:2 * n1] - np1, idx[:n1]) and torch.equal(out[n1:2 * n1], out[:n1])
    assert not torch.equal(idx[2 * n1:3 * n1] - 2 * np1, idx[:n1])             # other copies drew other noise
    per_graph = ratio.view(V, 6).sum(1)
    torch.testing.assert_close(per_graph, torch.ones_like(per_graph), rtol=1e-4, atol=1e-4)
    assert far.shape == (V, 1) and bool((far > 0).all())


# ------------------------------------------------------------------------------------ configs[1], dataset preset
def test_generator_forward_parity_at_dataset_size(cuda):
    from building_gan_amd.synthetic import batch_sizes, make_batch
    cpu = make_batch("dataset", seed=3, graphs=4, ragged="jitter")
    sz = batch_sizes(cpu)
    Go, _, Gh, _ = _nets(cuda, seed=3)
    torch.manual_seed(3)
    noise, pz, vz = _noise(sz), torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32)
    gpu = copy.deepcopy(cpu).to(cuda)
    with torch.no_grad():
        ro = Go(cpu, pz, vz, noise)
        rh = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))          # fused graph-free launches
    _compare_generator_outputs(rh, ro, "dataset preset, no-grad path")
    rh = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))              # autograd-node path of the generator update
    _compare_generator_outputs([r.detach() if torch.is_tensor(r) else {k: v.detach() for k, v in r.items()} for r in rh],
                               ro, "dataset preset, autograd path")


def _flat_to_named(opt, module):
    """{parameter name: gradient} out of a FlatAdam's flat gradient buffer."""
    names = [n for n, _ in module.named_parameters()]
    return {n: opt.grad[off:off + p.numel()].view_as(p).detach().cpu().clone()
            for n, p, off in zip(names, opt.params, opt.offsets)}


@pytest.mark.parametrize("preset,graphs,aux", [("small", None, (0.0, 0.0)), ("dataset", 2, (0.0, 0.0)),
                                               ("small", None, (1.0, 1.0))])
@pytest.mark.timeout(600)         # (the near-kink explanation, when needed, reruns the oracle ~30 times: ~1 min)
def test_trainer_flat_gradients_match_oracle(cuda, preset, graphs, aux):
    """The composition the iteration tests could not see (fused critic_terms -> CriticLoss -> FlatAdam.collect, and
    the generator phase through the frozen critic): the flat buffers handed to Adam / the all-reduce against the
    oracle's .grad after d_loss.backward() and g_loss.backward() (reference train.py:98-100, 111-113), no
    optimizer step in between on either side."""
    # The generator's output is a one-hot DECISION per voxel (straight-through estimator, models.py:128-141): where the two
    # sides' fp32 rounding straddles a tie between two logits they pick different labels, and every gradient then
    # differs at O(1 / Nv) -- a property of the draw, not of the gradient arithmetic.  So the comparison runs on the
    # first of three seeds on which both sides decide alike (the decisions are compared exactly), and fails if none does.
    for seed in (61, 62, 63):
        if _flat_gradients_on_seed(cuda, preset, graphs, aux, seed):
            return
    raise AssertionError("the one-hot decisions of the two sides differ on every seed tried")


def _flat_gradients_on_seed(cuda, preset, graphs, aux, seed):
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.train import Trainer, global_counts
    from oracle.objectives import train_iteration
    cpu = make_batch(preset, seed=seed, graphs=graphs, ragged=True if preset == "small" else "jitter")
    sz = batch_sizes(cpu)
    Go, Do, _, _ = _nets(cuda, seed=seed)
    tr = Trainer(cuda, tr_weight=aux[0], far_weight=aux[1])       # auxiliary losses: reference losses.py:65-68
    tr.generator.load_state_dict(Go.state_dict())
    tr.discriminator.load_state_dict(Do.state_dict())
    torch.manual_seed(seed)
    noise, pz, vz, u = _noise(sz), torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32), torch.rand(sz["Nv"], 1)
    with torch.no_grad():
        decided = Go(cpu, pz, vz, noise)[0].round()
    # oracle: one iteration with lr = 0 optimizers, so .grad holds what backward() left and no weight moves
    og, od = torch.optim.SGD(Go.parameters(), lr=0.0), torch.optim.SGD(Do.parameters(), lr=0.0)
    so = train_iteration(Go, Do, og, od, cpu, detach_batch, 0, pz, vz, noise, u, tr_weight=aux[0], far_weight=aux[1])
    gpu = copy.deepcopy(cpu).to(cuda)
    counts = global_counts(gpu)
    gen, sh = tr._phase_d(gpu, counts, True, pz.to(cuda), vz.to(cuda), _to(noise, cuda), u.to(cuda))
    differ = int((gen[0].detach().round().cpu() != decided).any(1).sum())
    if differ:
        print("%s seed %d: %d of %d voxels decided differently (a tie within fp32 rounding): next seed" % (preset, seed, differ, sz["Nv"]))
        return False
    d_grads = _flat_to_named(tr.opt_d, tr.discriminator)
    sh.update(tr._phase_g(gpu, counts, gen))
    g_grads = _flat_to_named(tr.opt_g, tr.generator)
    torch.testing.assert_close(sh["gp"].cpu(), so["gp"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(sh["d_loss"].cpu(), so["d_loss"], rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(sh["g_loss"].cpu().view(-1), so["g_loss"].view(-1), rtol=1e-3, atol=1e-3)
    worst = {}
    for tag, grads, net in (("D", d_grads, Do), ("G", g_grads, Go)):
        ref = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        signal, minor, zero = _grad_error_classes(tag, grads, ref)
        if tag == "G" and _over_the_bars(signal, minor, zero):
            # An activation unit whose pre-activation lies within the two sides' forward agreement of its kink can take
            # the other slope on the other side: the forward barely moves, the gradient moves by that unit's whole
            # share (a per-unit share is ~1 / sqrt(#units) of a weight gradient, 5e-3 here).  Such a difference is
            # accepted only when it is EXPLAINED: the oracle is rerun with named near-kink units on the other side
            # and must then agree to the same bars.
            def run(flips=(), record=None):
                return _oracle_generator_gradients(Go, Do, cpu, detach_batch, pz, vz, noise, u, aux, flips, record)
            ref_k, flipped = _reference_with_kink_sides(list(ref), run, grads)
            print("%s seed %d: generator gradients explained by %d activation unit(s) within %.0e of the kink taking "
                  "the other side: %s" % (preset, seed, len(flipped), KINK_EPS, flipped))
            assert flipped, (tag, "over the bars and no near-kink unit explains it", _over_the_bars(signal, minor, zero))
            signal, minor, zero = _grad_error_classes(tag, grads, ref_k)
        assert signal, tag
        worst[tag] = {"signal": max(signal.items(), key=lambda kv: kv[1]),
                      "minor": max(minor.items(), key=lambda kv: kv[1][0]) if minor else None,
                      "zero": max(zero.items(), key=lambda kv: kv[1]) if zero else None}
        assert not _over_the_bars(signal, minor, zero), (tag, _over_the_bars(signal, minor, zero))
    if aux[0]:
        assert float(sh["tr_loss"]) > 0 and float(sh["far_loss"]) > 0
    print("%s aux=%s seed %d: worst flat-gradient errors vs the fp32 oracle %s" % (preset, aux, seed, worst))
    return True


def _grad_error_classes(tag, grads, ref):
    top = max(float(g.abs().max()) for g in ref.values())
    signal, minor, zero = {}, {}, {}
    for n, g in ref.items():
        own = float(g.abs().max())
        err = float((grads[n].double() - g.double()).abs().max())
        if n in STRUCTURALLY_ZERO[tag]:
            # additive terms of the linear critic shift D(real) and D(fake) alike: the exact d_loss gradient is zero
            # (the penalty differentiates D's input gradient, which no additive term enters), what both sides hold
            # is rounding noise -- measured against the network's largest gradient
            assert own <= 1e-5 * top, (tag, n, own / top)
            zero[n] = err / top
        elif own >= SIGNAL_SHARE * top:
            signal[n] = err / own                 # relative to the tensor's OWN largest gradient: north_star's 1e-4
        else:
            minor[n] = (err / own, err / top)     # small but real gradients: a looser relative bar, a tight absolute one
    return signal, minor, zero


def _over_the_bars(signal, minor, zero):
    bad = [(n, e) for n, e in signal.items() if e > SIGNAL_GRAD_BAR]
    bad += [(n, e) for n, e in minor.items() if e[0] > MINOR_GRAD_BAR or e[1] > ZERO_GRAD_BAR]
    bad += [(n, e) for n, e in zero.items() if e > ZERO_GRAD_BAR]
    return sorted(bad, key=lambda kv: -(kv[1][0] if isinstance(kv[1], tuple) else kv[1]))[:6]


KINK_EPS = 1e-5          # |pre-activation| below which a unit counts as "at the kink" (forward agreement is ~1e-6 .. 1e-5)
KINK_MAX_UNITS = 32


def _activations(net):
    return [m for m in net.modules() if isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU))]


def _oracle_generator_gradients(G, D, cpu, detach_batch, pz, vz, noise, u, aux, flips=(), record=None):
    """The oracle's generator phase alone (train.py:106-113; iteration index 5 of a schedule that skips the critic),
    with the activation units in ``flips`` = [(module index, call index, element index)] evaluated on the OTHER side of
    their kink (slope exchanged; the value moves by <= KINK_EPS).  ``record`` collects the near-kink units."""
    from oracle.objectives import train_iteration
    G = copy.deepcopy(G)
    acts, hooks = _activations(G), []
    for mi, m in enumerate(acts):
        calls = [0]
        slope = float(getattr(m, "negative_slope", 0.0))

        def hook(mod, inp, out, mi=mi, calls=calls, slope=slope):
            c = calls[0]
            calls[0] += 1
            x = inp[0]
            if record is not None:
                near = (x.detach().abs() < KINK_EPS).nonzero()
                record.extend(((mi, c, tuple(int(t) for t in i)), float(x.detach()[tuple(i)].abs())) for i in near)
            mine = [idx for (m2, c2, idx) in flips if m2 == mi and c2 == c]
            if not mine:
                return None
            out = out.clone()
            for idx in mine:
                out[idx] = torch.where(x[idx] > 0, slope * x[idx], x[idx])
            return out
        hooks.append(m.register_forward_hook(hook))
    og, od = torch.optim.SGD(G.parameters(), lr=0.0), torch.optim.SGD(D.parameters(), lr=0.0)
    train_iteration(G, D, og, od, cpu, detach_batch, 5, pz, vz, noise, u, n_critic_d=7, n_critic_g=5,
                    tr_weight=aux[0], far_weight=aux[1])
    for h in hooks:
        h.remove()
    for p in D.parameters():
        p.requires_grad = True
    return {n: p.grad.detach().clone() for n, p in G.named_parameters()}


def _reference_with_kink_sides(names, run, grads):
    """Which near-kink units, put on the other side, explain ``grads`` - oracle?  Each candidate's gradient change is
    measured alone, a least-squares fit picks the subset (coefficient above one half), and the oracle is then run
    ONCE with exactly that subset flipped: that run is the reference returned."""
    flat = lambda d: torch.cat([d[n].double().reshape(-1) for n in names])
    record = []
    g0 = flat(run(record=record))
    units = [u for u, _ in sorted(record, key=lambda r: r[1])[:KINK_MAX_UNITS]]
    if not units:
        return None, []
    deltas = torch.stack([flat(run(flips=[u])) - g0 for u in units], 1)             # [n_param, K]
    coef = torch.linalg.lstsq(deltas, (flat(grads) - g0).view(-1, 1)).solution.view(-1)
    chosen = [u for u, c in zip(units, coef.tolist()) if c > 0.5]
    # (a unit whose flip barely moves any gradient gets an ill-determined coefficient; it does not matter which side it is
    # put on -- the run below with the chosen subset is what the comparison then rests on)
    if not chosen:
        return None, []
    return run(flips=chosen), chosen


# Measured on MI355X (tools/lab/grad_classes.py prints every tensor): tensors whose largest gradient
# is >= 1e-2 of their network's largest agree with the oracle to 2.0e-5 (D) / 1.5e-5 (G) of their own maximum at dataset
# size and 2.9e-6 / 3.2e-6 on the small preset; smaller ones to 9.5e-5 of their own and 8.5e-7 of the network's maximum;
# the critic's bias tensors hold 1e-7 of the network's maximum on both sides (exactly zero in exact arithmetic).
SIGNAL_SHARE = 1e-2
SIGNAL_GRAD_BAR = 1e-4        # north_star's fp32 tolerance, relative to the tensor's own largest gradient
MINOR_GRAD_BAR = 5e-4         # tensors below SIGNAL_SHARE: relative to their own maximum ..
ZERO_GRAD_BAR = 1e-5          # .. and, like the structurally zero ones, relative to the network's largest gradient
STRUCTURALLY_ZERO = {
    "D": {"voxelGNN_D.feature_enc.0.bias", "voxelGNN_D.label_enc.0.bias", "voxelGNN_D.block.messageMLP.0.bias",
          "voxelGNN_D.block.updateMLP.0.bias", "building_dec.0.bias", "building_dec.1.bias", "story_dec.0.bias",
          "story_dec.1.bias"},
    "G": set(),
}


def test_program_ratio_backward_parity(cuda):
    """get_program_ratio (reference util_graph.py:21-41) differentiated: gradients of a weighted sum of its three
    outputs with respect to the attention and the mask against the oracle's autograd."""
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.util_graph import get_program_ratio as h
    from oracle.pointer import get_program_ratio as o
    cpu = make_batch("small", seed=15, ragged=True)
    sz = batch_sizes(cpu)
    gpu = copy.deepcopy(cpu).to(cuda)
    torch.manual_seed(15)
    att, mask = torch.rand(sz["Ec"], 1), torch.rand(sz["Nv"], 1)
    w = [torch.rand(6 * sz["B"], 1), torch.rand(sz["Np"], 1), torch.rand(sz["B"], 1)]
    ao, mo = att.clone().requires_grad_(True), mask.clone().requires_grad_(True)
    ah, mh = att.to(cuda).requires_grad_(True), mask.to(cuda).requires_grad_(True)
    ro, rh = o(cpu, ao, mo, 6), h(gpu, ah, mh, 6)
    for a, b in zip(rh, ro):
        torch.testing.assert_close(a.detach().cpu(), b.detach(), rtol=1e-5, atol=1e-7)
    sum((r * wi).sum() for r, wi in zip(ro, w)).backward()
    sum((r * wi.to(cuda)).sum() for r, wi in zip(rh, w)).backward()
    assert _rel(ah.grad.cpu(), ao.grad) <= 1e-5 and _rel(mh.grad.cpu(), mo.grad) <= 1e-5


def _batch64(b):
    b = copy.deepcopy(b)
    for k in b.keys:
        v = getattr(b, k)
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(b, k, v.double())
    return b


@pytest.mark.parametrize("preset,graphs", [("small", None), ("dataset", 2)])
def test_generator_backward_and_penalty_against_float64(cuda, preset, graphs):
    """The float64 treatment the discriminator already had (VERDICT weak 3): the generator's parameter gradients
    through the critic and the gradient penalty's double backward, HIP (fp32) and oracle (fp32) each measured
    against the oracle run in float64 on identical inputs and noise.  The HIP path has to be as close to float64
    as the fp32 oracle is: per tensor, error <= max(GRAD64_BAR, 2 x the fp32 oracle's own error) of the tensor's scale."""
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.losses import VolumetricDesignLoss_G, compute_gradient_penalty as gp_h
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from oracle.objectives import compute_gradient_penalty as gp_o, g_loss_fn
    cpu = make_batch(preset, seed=71, graphs=graphs, ragged=True if preset == "small" else "jitter")
    sz = batch_sizes(cpu)
    Go, Do, Gh, Dh = _nets(cuda, seed=71)
    G64, D64 = copy.deepcopy(Go).double(), copy.deepcopy(Do).double()
    cpu64 = _batch64(cpu)
    torch.manual_seed(71)
    noise, pz, vz, u = _noise(sz), torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32), torch.rand(sz["Nv"], 1)
    noise64 = [(a.double(), b.double()) for a, b in noise]
    gpu = copy.deepcopy(cpu).to(cuda)
    # ---- generator backward through the critic
    out64, _, mask64, att64, idx64 = G64(cpu64, pz.double(), vz.double(), noise64)
    g_loss_fn(D64(cpu64, out64), cpu64, att64["hard"], mask64["hard"], 6)[0].backward()
    out_o, _, mask_o, att_o, idx_o = Go(cpu, pz, vz, noise)
    g_loss_fn(Do(cpu, out_o), cpu, att_o["hard"], mask_o["hard"], 6)[0].backward()
    out_h, _, mask_h, att_h, idx_h = Gh(gpu, pz.to(cuda), vz.to(cuda), _to(noise, cuda))
    VolumetricDesignLoss_G()(Dh(gpu, out_h), gpu, att_h["hard"], mask_h["hard"], 6)[0].backward()
    same = torch.equal(idx_h.cpu(), idx64) and torch.equal(idx_o, idx64)
    eg_h = {n: _rel(p.grad.cpu(), q.grad) for (n, p), (_, q) in zip(Gh.named_parameters(), G64.named_parameters())}
    eg_o = {n: _rel(p.grad, q.grad) for (n, p), (_, q) in zip(Go.named_parameters(), G64.named_parameters())}
    # ---- gradient penalty: double backward
    soft = torch.rand(sz["Nv"], 6).softmax(-1) * (torch.rand(sz["Nv"], 1) > 0.3).float()
    for net in (D64, Do, Dh):
        net.zero_grad()
    gp64 = gp_o(D64, detach_batch(cpu64), cpu64.voxel_label, soft.double(), u.double())[0]
    gp64.backward()
    gpo = gp_o(Do, detach_batch(cpu), cpu.voxel_label, soft, u)[0]
    gpo.backward()
    gph = gp_h(Dh, detach_batch(gpu), gpu.voxel_label, soft.to(cuda), u.to(cuda))[0]
    gph.backward()
    ed_h, ed_o = {}, {}
    top = max(float(q.grad.abs().max()) for q in D64.parameters() if q.grad is not None)
    for (n, p), (_, po), (_, q) in zip(Dh.named_parameters(), Do.named_parameters(), D64.named_parameters()):
        if q.grad is None or float(q.grad.abs().max()) == 0.0:
            # additive terms (biases) do not move d D / d sample: no gradient in the reference either
            assert p.grad is None or float(p.grad.abs().max()) <= 1e-6 * top, n
            continue
        ed_h[n], ed_o[n] = _rel(p.grad.cpu(), q.grad), _rel(po.grad, q.grad)
    w = lambda d: max(d.items(), key=lambda kv: kv[1])
    print("%s (identical pointers: %s): generator gradients vs float64 -- HIP worst %s, fp32 oracle worst %s; "
          "penalty %.3e (HIP) %.3e (oracle) rel.; penalty gradients -- HIP worst %s, fp32 oracle worst %s"
          % (preset, same, w(eg_h), w(eg_o), abs(float(gph) - float(gp64)) / float(gp64),
             abs(float(gpo) - float(gp64)) / float(gp64), w(ed_h), w(ed_o)))
    assert abs(float(gph) - float(gp64)) <= 1e-5 * float(gp64)
    # as close to float64 as the fp32 oracle is, or inside the bar (<= 10x the measured 3e-6): where fp32 and
    # float64 pick different arg-max voxels in the critic's pooling (dataset size: both fp32 paths then sit 3e-2
    # from float64, and 1e-5 from each other) only the first criterion is meaningful
    for n in ed_h:
        assert ed_h[n] <= max(GRAD64_BAR["penalty"], 2 * ed_o[n]), ("penalty", n, ed_h[n], ed_o[n])
    for n in eg_h:
        assert eg_h[n] <= max(GRAD64_BAR["generator"], 2 * eg_o[n]), ("generator", n, eg_h[n], eg_o[n])
    # and the two fp32 paths against each other -- meaningful where the fp32 oracle itself agrees with float64; where
    # it picked another arg-max voxel than float64 did (its own error jumps to ~3e-2 at dataset size) the HIP path may
    # side with float64 instead (it does in the Chebyshev basis: 5e-4 from float64), so the bound is the oracle's own error
    for (n, p), (_, po) in zip(Gh.named_parameters(), Go.named_parameters()):
        bound = 1e-3 if eg_o[n] <= 1e-3 else 2 * eg_o[n]
        assert _rel(p.grad.cpu(), po.grad) <= bound, ("generator, HIP vs fp32 oracle", n, eg_h[n], eg_o[n])


GRAD64_BAR = {"generator": 3e-5, "penalty": 3e-5}
