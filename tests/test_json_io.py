"""Raw JSON schema round trip: synthetic design -> three JSON files ->
reader restating GraphConstructor -> identical tensors; and the output writer of util_eval.save_output."""
import json
import os

import numpy as np
import torch


def _graphs():
    from building_gan_amd.synthetic import make_graph
    rng = np.random.default_rng(11)
    return [(make_graph(rng, 3, 4, 3, [3, 1, 7], data_id=0), 4), (make_graph(rng, 1, 2, 2, [2], data_id=1), 2)]


def test_reader_reproduces_every_tensor(tmp_path):
    from building_gan_amd.graph import GRAPH_KEYS
    from building_gan_amd.json_io import load_graph_jsons, write_raw_jsons
    for g, gx in _graphs():
        write_raw_jsons(g, gx, str(tmp_path))
        r = load_graph_jsons(int(g.data_id_str), str(tmp_path))
        for k in GRAPH_KEYS:
            a, b = g[k], r[k]
            if torch.is_tensor(a):
                assert a.dtype == b.dtype and a.shape == b.shape, k
                if a.is_floating_point():
                    torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7, msg=k)
                else:
                    assert torch.equal(a, b), k
            else:
                assert a == b, k


def test_cross_edges_follow_the_reference_construction():
    # hand case: story 0 has program nodes {0,1} and 2 voxels, story 1 has node {2} and 1 voxel
    from building_gan_amd.json_io import jsons_to_graph
    local = {"node": [{"floor": 0, "type": 1, "type_id": 0, "neighbors": [[0, 2, 0]]},
                      {"floor": 0, "type": 2, "type_id": 0, "neighbors": [[0, 1, 0], [1, 1, 0]]},
                      {"floor": 1, "type": 1, "type_id": 0, "neighbors": [[0, 2, 0]]}]}
    glob = {"far": 2.5, "global_node": [{"type": 1, "proportion": 0.75}, {"type": 2, "proportion": 0.25}]}
    vox = {"voxel_node": [
        {"location": [0, 0, 0], "type": 2, "type_id": 0, "dimension": [4, 3, 3], "coordinate": [0, 0, 0], "weight": 1.0, "neighbors": [[0, 0, 1], [1, 0, 0]]},
        {"location": [0, 0, 1], "type": -1, "type_id": 0, "dimension": [4, 3, 3], "coordinate": [4, 0, 0], "weight": 0.5, "neighbors": [[0, 0, 0]]},
        {"location": [1, 0, 0], "type": 1, "type_id": 0, "dimension": [4, 3, 3], "coordinate": [0, 0, 3], "weight": 1.0, "neighbors": [[0, 0, 0]]}]}
    g = jsons_to_graph(glob, local, vox, "000007")
    assert g.cross_edge_program_index_select.tolist() == [0, 1, 0, 1, 2]
    assert g.cross_edge_voxel_index_select.tolist() == [0, 0, 1, 1, 2]
    assert g.voxel_label_program_node_index.tolist() == [1, -1, 2]
    assert g.program_edge.tolist() == [[0, 1, 1, 2], [1, 0, 2, 1]]
    assert g.neg_same_story_program_edge.shape == (2, 0) and g.neg_diff_story_program_edge.tolist() == [[0, 2], [2, 0]]
    assert g.program_target_ratio.tolist() == [0.0, 0.75, 0.25, 0.0, 0.0, 0.0] and g.FAR.tolist() == [2.5]
    assert g.voxel_label.tolist() == [[0, 0, 1, 0, 0, 0], [0] * 6, [0, 1, 0, 0, 0, 0]]
    assert g.voxel_edge.tolist() == [[0, 0, 1, 2], [1, 2, 0, 0]] and g.voxel_edge_mask.tolist() == [1.0, 0.0, 1.0, 0.0]
    assert g.voxel_projection_cluster.tolist() == [0, 1, 0]
    np.testing.assert_allclose(g.voxel_feature[1].numpy(), [0.04, 0.03, 0.03, 0.04, 0, 0, 0.5], rtol=1e-6)


def test_save_output_writes_labels_ratios_and_far(tmp_path):
    from building_gan_amd.graph import Batch
    from building_gan_amd.json_io import save_output, write_raw_jsons
    raw, outd = str(tmp_path / "raw"), str(tmp_path / "out")
    graphs = _graphs()
    for g, gx in graphs:
        write_raw_jsons(g, gx, raw)
    batch = Batch.from_data_list([g for g, _ in graphs])
    nv, np_ = batch.voxel_feature.shape[0], batch.program_class_feature.shape[0]
    torch.manual_seed(0)
    pid = batch.voxel_label_program_node_index.clone()
    void = pid < 0
    first_node = torch.tensor([0 if i < graphs[0][0].voxel_feature.shape[0] else graphs[0][0].program_class_feature.shape[0] for i in range(nv)])
    pid = torch.where(void, first_node, pid)
    out = batch.voxel_label.clone()                       # painted voxels carry an exact 1.0, void ones all zeros
    cw = torch.rand(batch.program_target_ratio.shape[0], 1)
    pw = torch.rand(np_, 1)
    far = torch.tensor([[1.5], [2.5]])
    res = save_output(2, batch, cw, pw, far, pid, out, None, raw, outd)
    assert len(res) == 2
    for gi, (g, gx) in enumerate(graphs):
        ids = g.data_id_str
        gg = json.load(open(os.path.join(outd, "global_graph_data", "graph_global_%s.json" % ids)))
        lg = json.load(open(os.path.join(outd, "local_graph_data", "graph_local_%s.json" % ids)))
        vg = json.load(open(os.path.join(outd, "voxel_data", "voxel_%s.json" % ids)))
        assert abs(gg["new_far"] - float(far[gi])) < 1e-6
        for n in gg["global_node"]:
            assert abs(n["new_proportion"] - float(cw[6 * gi + n["type"]])) < 1e-6
        raw_v = json.load(open(os.path.join(raw, "voxel_data", "voxel_%s.json" % ids)))
        for a, b in zip(vg["voxel_node"], raw_v["voxel_node"]):
            assert a["type"] == b["type"] and a["type_id"] == b["type_id"]      # labels painted back unchanged
        assert all("region_far" in n for n in lg["node"])


def test_draw_z_skips_the_noise_the_generator_never_reads():
    """reference util_eval.py:81-87 draws voxel_z although Generator builds its voxel GNN with noise_dim 0
    (models.py:239); the mirror draws it only for a generator that reads it."""
    import types
    import torch
    from building_gan_amd.models import Generator
    from building_gan_amd.util_eval import draw_z
    g = types.SimpleNamespace(program_class_feature=torch.zeros(5, 6), voxel_feature=torch.zeros(9, 7))
    G = Generator(7, 7, 16, 4, 1, 1)
    pz, vz = draw_z(g, 4, torch.device("cpu"), 1.0, G)
    assert tuple(pz.shape) == (5, 4) and vz is None
    pz, vz = draw_z(g, 4, torch.device("cpu"))                 # no generator given: both, as the reference
    assert tuple(pz.shape) == (5, 4) and tuple(vz.shape) == (9, 4)
