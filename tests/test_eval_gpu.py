"""Inference-side helpers on the GPU against the oracle."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset,ragged", [("small", True), ("dataset", "jitter"), ("tiny", False)])
def test_check_connectivity_matches_reference_semantics(cuda, preset, ragged):
    """util_eval.check_connectivity (bg_connectivity: a code table over program-node pairs and three sorted lists, one
    read-back) against the oracle's torch.unique form: identical edge lists in identical order; the bitmap form in torch
    that it replaces gives the same; the reference's else-branch when no voxel edge survives the mask."""
    from building_gan_amd import util_eval
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.util_eval import check_connectivity as h
    from oracle.evaluation import check_connectivity as o
    cpu = make_batch(preset, seed=41, ragged=ragged)
    sz = batch_sizes(cpu)
    torch.manual_seed(41)
    # a plausible generator output: every voxel points at a program node of its own story
    cev, cep = cpu.cross_edge_voxel_index_select, cpu.cross_edge_program_index_select
    first = torch.zeros(sz["Nv"], dtype=torch.long).scatter_reduce(0, cev, torch.arange(sz["Ec"]), "amin", include_self=False)
    count = torch.bincount(cev, minlength=sz["Nv"])
    pick = first + (torch.rand(sz["Nv"]) * count).long()
    idx = cep[pick]
    mask = (torch.rand(sz["Nv"], 1) > 0.3).float()
    ro = o(cpu, idx, mask)
    gpu = copy.deepcopy(cpu).to(cuda)
    rh = h(gpu, idx.to(cuda), mask.to(cuda))
    for a, b in zip(rh, ro):
        assert a.dtype == b.dtype and torch.equal(a.cpu(), b)
    orig = util_eval._check_connectivity_hip
    try:                                                   # the torch bitmap form (inputs the entry point declines)
        util_eval._check_connectivity_hip = lambda *a: None
        rt = h(gpu, idx.to(cuda), mask.to(cuda))
    finally:
        util_eval._check_connectivity_hip = orig
    for a, b in zip(rh, rt):
        assert torch.equal(a, b)
    # every voxel unmasked, and every voxel assigned to ONE program node (a single generated pair, a self-loop)
    ones = torch.ones(sz["Nv"], 1)
    for ix in (idx, torch.zeros_like(idx)):
        ro1 = o(cpu, ix, ones)
        rh1 = h(gpu, ix.to(cuda), ones.to(cuda))
        for a, b in zip(rh1, ro1):
            assert torch.equal(a.cpu(), b)
    # nothing unmasked -> the reference's else branch
    rh = h(gpu, idx.to(cuda), torch.zeros(sz["Nv"], 1, device=cuda))
    assert rh[0].shape[1] == 0 and torch.equal(rh[1].cpu(), cpu.program_edge) and rh[2].shape[1] == 0


def test_evaluate_and_variation_sampling_run(cuda):
    from building_gan_amd import models as hm
    from building_gan_amd.graph import Batch
    from building_gan_amd.synthetic import make_batch, make_graph
    from building_gan_amd.util_eval import evaluate, generate_multiple_outputs_from_data
    import numpy as np
    torch.manual_seed(0)
    G = hm.Generator(7, 7, 128, 32, 4, 12, cuda).to(cuda)
    loader = [make_batch("small", seed=s, ragged=True) for s in (1, 2)]
    torch.manual_seed(5)
    acc, res = evaluate(loader, G, None, None, None, [cuda], number_of_batches=2)
    assert 0.0 <= acc <= 1.0 and len(res) == 2
    # the second design's plan was prefetched beside the first design's forward (side stream): the same results as
    # designs evaluated one at a time, each building its own plan
    torch.manual_seed(5)
    single = [evaluate([make_batch("small", seed=s, ragged=True)], G, None, None, None, [cuda], number_of_batches=1)[1][0]
              for s in (1, 2)]
    for a, b in zip(res, single):
        for k in ("out", "max_out_program_index", "class_weights", "FAR", "inter_edges", "missing_edges", "gen_edges"):
            assert torch.equal(a[k], b[k]), k
    assert len(evaluate(loader, G, None, None, None, [cuda], number_of_batches=1)[1]) == 1
    assert res[0]["out"].shape[1] == 6 and res[0]["FAR"].shape[0] == loader[0].FAR.shape[0]
    data = make_graph(np.random.default_rng(3), 3, 4, 3, [3, 2, 4])
    out, idx, cw, pw, far = generate_multiple_outputs_from_data(data, 25, G, None, None, None, [cuda])
    nv = data.voxel_feature.shape[0]
    assert out.shape == (25 * nv, 6) and far.shape == (25, 1)
    # different noise per copy -> the 25 variations are not all identical
    copies = out.view(25, nv, 6)
    assert not all(torch.equal(copies[0], copies[k]) for k in range(1, 25))


def test_large_building_needs_a_longer_position_table(cuda):
    """BASELINE configs[4]: a 32-story building exceeds the reference's 20-row table (models.py:13,192)."""
    from building_gan_amd import models as hm
    from building_gan_amd.graph import Batch
    from building_gan_amd.synthetic import make_graph
    import numpy as np
    data = make_graph(np.random.default_rng(9), 32, 16, 16, [4] * 32)
    batch = Batch.from_data_list([data, data]).to(cuda)
    torch.manual_seed(1)
    G20 = hm.Generator(7, 7, 128, 32, 4, 12, cuda).to(cuda)
    z = (torch.rand(batch.program_class_feature.shape[0], 32, device=cuda), torch.rand(batch.voxel_feature.shape[0], 32, device=cuda))
    with pytest.raises(IndexError):
        G20(batch, *z)
    G32 = hm.Generator(7, 7, 128, 32, 4, 12, cuda, max_len=32).to(cuda)
    with torch.no_grad():
        out, soft_out, mask, att, idx = G32(batch, *z)
    assert out.shape == (batch.voxel_feature.shape[0], 6) and torch.isfinite(soft_out).all()
    lo, hi = batch.__slices__["program_class_feature"][1], batch.__slices__["program_class_feature"][2]
    nv = data.voxel_feature.shape[0]
    assert int(idx[nv:].min()) >= lo and int(idx[nv:].max()) < hi      # second copy points into its own program nodes


def test_trained_generator_round_trips_through_a_checkpoint(cuda, tmp_path):
    """Trainer.fit (the reference's epoch loop + checkpoint writer, train.py:73-75,133-141) -> <epoch>.ckpt ->
    inference.load_generator (weights_only=True): the reloaded generator reproduces the trained one bit for bit,
    although the trained parameters are views of the optimizer's flat buffer."""
    import os
    from building_gan_amd.inference import load_generator
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.train import Trainer
    loader = [make_batch("small", seed=s, ragged=True) for s in (5, 6)]
    torch.manual_seed(0)
    tr = Trainer(cuda)
    before = tr.opt_g.flat.clone()
    stats = tr.fit(loader, 2, model_dir=str(tmp_path), eval_period=1)
    assert torch.isfinite(stats["d_loss"]).all() and not torch.equal(tr.opt_g.flat, before)
    assert sorted(os.listdir(tmp_path)) == ["0.ckpt", "1.ckpt"]
    state = torch.load(str(tmp_path / "1.ckpt"), weights_only=True)
    assert set(state) == set(tr.generator.state_dict()) and all(v.is_contiguous() for v in state.values())
    assert sum(v.numel() for k, v in state.items() if k != "voxelGNN.pe") == 228418
    gen = load_generator(str(tmp_path / "1.ckpt"), cuda)
    batch = make_batch("small", seed=7, ragged=True).to(cuda)
    sz = batch_sizes(batch)
    noise = [(torch.empty(sz["Nv"], 2, device=cuda).exponential_(), torch.empty(sz["Ec"], device=cuda).exponential_())
             for _ in range(6)]
    pz = torch.rand(sz["Np"], 32, device=cuda)
    with torch.no_grad():
        a = tr.generator(batch, pz, None, noise)
        b = gen(batch, pz, None, noise)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])


def test_inference_driver_with_released_weights(cuda, tmp_path):
    """checkpoint file -> generator (strict=False) -> raw JSON designs -> result JSONs + 5 variations."""
    import json
    import os
    import numpy as np
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_util import load_golden
    from building_gan_amd.inference import run
    from building_gan_amd.json_io import write_raw_jsons
    from building_gan_amd.synthetic import make_graph
    weights, _, _ = load_golden()
    weights["programGNN.encNorm.weight"] = torch.ones(1)        # an extra key, as in the released file
    ckpt = str(tmp_path / "70.ckpt")
    torch.save(weights, ckpt)
    raw = str(tmp_path / "raw")
    rng = np.random.default_rng(2)
    for i, (f, x, y) in enumerate([(3, 4, 3), (2, 3, 3), (4, 2, 2)]):
        write_raw_jsons(make_graph(rng, f, x, y, [3] * f, data_id=i), x, raw)
    acc = run(ckpt, raw, str(tmp_path / "out"), [0, 1, 2], cuda, batch_size=2, variation_num=5, variation_ids=[1])
    assert 0.0 <= acc <= 1.0
    for i in range(3):
        v = json.load(open(tmp_path / "out" / "output" / "voxel_data" / ("voxel_%06d.json" % i)))
        assert all(n["type"] in (-1, 0, 1, 2, 3, 4, 5) for n in v["voxel_node"])
        g = json.load(open(tmp_path / "out" / "output" / "global_graph_data" / ("graph_global_%06d.json" % i)))
        assert "new_far" in g and all("new_proportion" in n for n in g["global_node"])
    assert len(os.listdir(tmp_path / "out" / "var_output1" / "voxel_data")) == 5
