"""Raw JSON <-> tensors: the on-disk schema of the reference data set.

  * reading: restates /root/reference/Data/GraphConstructor.py:30-219 (three files per design:
    ``global_graph_data/graph_global_XXXXXX.json``, ``local_graph_data/graph_local_XXXXXX.json``,
    ``voxel_data/voxel_XXXXXX.json``) and produces a ``VolumeDesignGraph`` with the same tensors;
  * writing: restates ``save_output`` of /root/reference/util_eval.py:9-70 (copies the raw JSONs
    and fills in ``new_far`` / ``new_proportion`` / ``region_far`` / voxel ``type`` + ``type_id``);
  * ``graph_to_jsons`` is the inverse of the reader, used to materialise synthetic designs in
    the raw schema (the real data set is not redistributable offline).
Host-side only (json + torch CPU tensors); nothing here is on the GPU hot path.
"""
from __future__ import annotations

import json
import os
from collections import Counter
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .graph import VolumeDesignGraph

NUMBER_OF_CLASS = 6            # GraphConstructor.py:9
DIMENSION_NORM_FACTOR = 100    # GraphConstructor.py:10
GLOBAL_PREFIX, LOCAL_PREFIX, VOXEL_PREFIX = "graph_global_", "graph_local_", "voxel_"
DATA_ID_LENGTH = 6


def _paths(data_dir: str, data_id_str: str) -> Tuple[str, str, str]:
    return (os.path.join(data_dir, "global_graph_data", GLOBAL_PREFIX + data_id_str + ".json"),
            os.path.join(data_dir, "local_graph_data", LOCAL_PREFIX + data_id_str + ".json"),
            os.path.join(data_dir, "voxel_data", VOXEL_PREFIX + data_id_str + ".json"))


def _one_hot(t: int) -> List[int]:
    assert t < NUMBER_OF_CLASS
    return [1 if i == t else 0 for i in range(NUMBER_OF_CLASS)]


def jsons_to_graph(global_graph: dict, local_graph: dict, raw_voxel_graph: dict, data_id_str: str) -> VolumeDesignGraph:
    # ---- program graph (GraphConstructor.py:163-219)
    nodes = local_graph["node"]
    key_to_id = {(n["floor"], n["type"], n["type_id"]): i for i, n in enumerate(nodes)}
    floors = [n["floor"] for n in nodes]
    types = [n["type"] for n in nodes]
    src, dst = [], []
    adj = set()
    for i, n in enumerate(nodes):
        for nb in n["neighbors"]:
            j = key_to_id[tuple(nb)]
            src.append(i)
            dst.append(j)
            adj.add((i, j))
    same, diff = [[], []], [[], []]
    for i in range(len(nodes)):
        for j in range(i + 1, len(nodes)):
            if (i, j) in adj:
                continue
            tgt = same if floors[i] == floors[j] else diff
            tgt[0] += [i, j]
            tgt[1] += [j, i]
    ratio = [0.0] * NUMBER_OF_CLASS
    for c in global_graph["global_node"]:
        ratio[c["type"]] = c["proportion"]

    # ---- voxel graph (GraphConstructor.py:120-160)
    vnodes = raw_voxel_graph["voxel_node"]
    loc_to_id = {tuple(n["location"]): i for i, n in enumerate(vnodes)}
    v_floor = [n["location"][0] for n in vnodes]
    feature = [[x / DIMENSION_NORM_FACTOR for x in n["dimension"] + n["coordinate"]] + [n["weight"]] for n in vnodes]
    proj_ids: Dict[Tuple[int, int], int] = {}
    proj = []
    for n in vnodes:
        proj.append(proj_ids.setdefault((n["location"][2], n["location"][1]), len(proj_ids)))
    e_src, e_dst = [], []
    for i, n in enumerate(vnodes):
        for nb in n["neighbors"]:
            e_src.append(i)
            e_dst.append(loc_to_id[tuple(nb)])

    # ---- cross edges: every voxel of a story x every program node of that story (GraphConstructor.py:71-93)
    groups: List[List[int]] = []
    for i, f in enumerate(floors):
        if i > 0 and f == floors[i - 1]:
            groups[-1].append(i)
        else:
            groups.append([i])
    per_story = Counter(v_floor)
    cep, cev, vid = [], [], 0
    for story_id, group in enumerate(groups):
        nv = per_story[story_id]
        cep += group * nv
        for _ in range(nv):
            cev += [vid] * len(group)
            vid += 1
    assert len(cep) == len(cev)

    label_node = []
    for i, n in enumerate(vnodes):
        pid = key_to_id.get((v_floor[i], n["type"], n["type_id"]), -1)
        if pid != -1:
            assert floors[pid] == v_floor[i]
        label_node.append(pid)
    edge_mask = [1.0 if v_floor[a] == v_floor[b] else 0.0 for a, b in zip(e_src, e_dst)]

    lt = lambda x: torch.tensor(x, dtype=torch.long)
    ft = lambda x: torch.tensor(x, dtype=torch.float32)
    return VolumeDesignGraph(
        data_id_str=data_id_str,
        program_floor_cluster=lt(floors), story_level_feature=ft(floors),
        program_class_feature=ft([_one_hot(t) for t in types]).view(-1, NUMBER_OF_CLASS),
        program_edge=lt([src, dst]).view(2, -1),
        neg_same_story_program_edge=lt(same).view(2, -1), neg_diff_story_program_edge=lt(diff).view(2, -1),
        program_class_cluster=lt(types), program_target_ratio=ft(ratio), FAR=ft([global_graph["far"]]),
        voxel_floor_cluster=lt(v_floor), voxel_projection_cluster=lt(proj),
        voxel_feature=ft(feature).view(-1, 7), voxel_bool=ft([0 if n["type"] < 0 else 1 for n in vnodes]),
        voxel_label=ft([_one_hot(n["type"]) for n in vnodes]).view(-1, NUMBER_OF_CLASS),
        voxel_edge=lt([e_src, e_dst]).view(2, -1), voxel_edge_mask=ft(edge_mask),
        cross_edge_program_index_select=lt(cep), cross_edge_voxel_index_select=lt(cev),
        voxel_label_program_node_index=lt(label_node),
    )


def load_graph_jsons(data_id, data_dir: str) -> VolumeDesignGraph:
    """reference GraphConstructor.load_graph_jsons (GraphConstructor.py:30-42)."""
    data_id_str = str(data_id).zfill(DATA_ID_LENGTH)
    gp, lp, vp = _paths(data_dir, data_id_str)
    with open(gp) as f:
        global_graph = json.load(f)
    with open(lp) as f:
        local_graph = json.load(f)
    with open(vp) as f:
        voxel_graph = json.load(f)
    return jsons_to_graph(global_graph, local_graph, voxel_graph, data_id_str)


def graph_to_jsons(graph: VolumeDesignGraph, grid_x: int) -> Tuple[dict, dict, dict]:
    """Inverse of the reader for grid-shaped designs whose voxels are ordered (story, y, x)."""
    floors = graph.program_floor_cluster.tolist()
    types = graph.program_class_cluster.tolist()
    seen: Dict[Tuple[int, int], int] = {}
    type_ids = []
    for f, t in zip(floors, types):
        type_ids.append(seen.get((f, t), 0))
        seen[(f, t)] = type_ids[-1] + 1
    keys = [[f, t, k] for f, t, k in zip(floors, types, type_ids)]
    nbrs: List[List[List[int]]] = [[] for _ in floors]
    for a, b in graph.program_edge.t().tolist():
        nbrs[a].append(keys[b])
    local_graph = {"node": [{"floor": f, "type": t, "type_id": k, "neighbors": nb, "region_far": 0.0, "center": [0, 0, 0]}
                            for (f, t, k), nb in zip(keys, nbrs)]}
    ratio = graph.program_target_ratio.tolist()
    global_graph = {"far": float(graph.FAR[0]), "site_area": 0.0,
                    "global_node": [{"type": t, "proportion": ratio[t], "connections": []}
                                    for t in range(NUMBER_OF_CLASS) if ratio[t] > 0]}
    vf = graph.voxel_feature.double()
    vfloor = graph.voxel_floor_cluster.tolist()
    proj = graph.voxel_projection_cluster.tolist()
    locs = [[fl, p // grid_x, p % grid_x] for fl, p in zip(vfloor, proj)]
    vn: List[List[List[int]]] = [[] for _ in locs]
    for a, b in graph.voxel_edge.t().tolist():
        vn[a].append(locs[b])
    label_node = graph.voxel_label_program_node_index.tolist()
    voxel_nodes = []
    for i, loc in enumerate(locs):
        pid = label_node[i]
        voxel_nodes.append({
            "location": loc, "type": types[pid] if pid >= 0 else -1, "type_id": type_ids[pid] if pid >= 0 else 0,
            "dimension": [round(x, 6) for x in (vf[i, 0:3] * DIMENSION_NORM_FACTOR).tolist()],
            "coordinate": [round(x, 6) for x in (vf[i, 3:6] * DIMENSION_NORM_FACTOR).tolist()],
            "weight": float(vf[i, 6]), "neighbors": vn[i]})
    return global_graph, local_graph, {"voxel_node": voxel_nodes}


def write_raw_jsons(graph: VolumeDesignGraph, grid_x: int, data_dir: str, data_id: Optional[int] = None) -> str:
    data_id_str = graph.data_id_str if data_id is None else str(data_id).zfill(DATA_ID_LENGTH)
    for sub in ("global_graph_data", "local_graph_data", "voxel_data"):
        os.makedirs(os.path.join(data_dir, sub), exist_ok=True)
    for path, obj in zip(_paths(data_dir, data_id_str), graph_to_jsons(graph, grid_x)):
        with open(path, "w") as f:
            json.dump(obj, f)
    return data_id_str


def _graph_ranges(batch, key: str) -> List[Tuple[int, int]]:
    sl = batch.__slices__[key]
    return [(sl[i], sl[i + 1]) for i in range(len(sl) - 1)]


def save_output(batch_size, batch, class_weights, program_weights, FAR, max_out_program_index, out, follow_batch,
                raw_dir, output_dir, new_data_id_strs: Optional[Sequence] = None):
    """reference util_eval.save_output (util_eval.py:9-70): one (global, local, voxel) JSON triple per graph."""
    for sub in ("global_graph_data", "local_graph_data", "voxel_data"):
        os.makedirs(os.path.join(output_dir, sub), exist_ok=True)
    vr = _graph_ranges(batch, "voxel_feature")
    pr = _graph_ranges(batch, "program_class_feature")
    sr = _graph_ranges(batch, "program_target_ratio")
    out = out.detach().cpu()
    cw_all, pw_all = class_weights.detach().cpu().view(-1), program_weights.detach().cpu().view(-1)
    far_all, pid_all = FAR.detach().cpu().view(-1), max_out_program_index.detach().cpu()
    results = []
    for i in range(len(vr)):
        data_id_str = batch.data_id_str[i]
        new_id = data_id_str if new_data_id_strs is None else str(new_data_id_strs[i]).zfill(DATA_ID_LENGTH)
        gp, lp, vp = _paths(raw_dir, data_id_str)
        ogp, olp, ovp = _paths(output_dir, new_id)
        cw = cw_all[sr[i][0]:sr[i][1]]
        pw = pw_all[pr[i][0]:pr[i][1]]
        with open(gp) as f:
            global_graph = json.load(f)
        global_graph["new_far"] = float(far_all[i])
        for n in global_graph["global_node"]:
            n["new_proportion"] = float(cw[n["type"]])
        with open(ogp, "w") as f:
            json.dump(global_graph, f)
        with open(lp) as f:
            local_graph = json.load(f)
        lookup = {}
        for k, (n, c) in enumerate(zip(local_graph["node"], pw.tolist())):
            n["region_far"] = float(c)
            lookup[k] = (n["type"], n["type_id"])
        with open(olp, "w") as f:
            json.dump(local_graph, f)
        with open(vp) as f:
            voxel_graph = json.load(f)
        labels = out[vr[i][0]:vr[i][1]].tolist()
        pids = pid_all[vr[i][0]:vr[i][1]].tolist()
        for n, label, pid in zip(voxel_graph["voxel_node"], labels, pids):
            t, tid = lookup[pid - pr[i][0]]
            painted = 1.0 in label              # the reference's test: a straight-through 1 that is exactly 1.0
            n["type"] = t if painted else -1
            n["type_id"] = tid if painted else 0
        with open(ovp, "w") as f:
            json.dump(voxel_graph, f)
        results.append([global_graph, local_graph, voxel_graph])
    return results
