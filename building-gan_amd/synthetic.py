"""Synthetic "6types-shaped" volumetric design graphs.

The reference data set is not redistributable offline, so benchmarks and tests use graphs that
follow the tensor layout produced by the reference's JSON -> tensor constructor:
  * voxel nodes, features and 6-neighbour edges ... /root/reference/Data/GraphConstructor.py:120-160
  * program nodes, classes, edges ................. /root/reference/Data/GraphConstructor.py:163-219
  * cross edges (complete bipartite per story) .... /root/reference/Data/GraphConstructor.py:71-93
  * constants (6 classes, /100 normalisation) ..... /root/reference/Data/GraphConstructor.py:9-10
Presets are the ones SURVEY.md section 8(d) names.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graph import Batch, VolumeDesignGraph, DEFAULT_FOLLOW_BATCH

NUMBER_OF_CLASS = 6
DIMENSION_NORM_FACTOR = 100.0

# name -> (graphs per batch, stories, grid X, grid Y)
PRESETS = {
    "tiny": (2, 2, 3, 2),
    "small": (4, 3, 5, 4),
    "dataset": (8, 10, 12, 12),      # Nv=11,520  Ev=62,976  Ec=46,080 (with 4 program nodes / story)
    "microbench": (8, 20, 33, 33),   # Nv=174,240 Ev=1,006,896
    "large": (1, 32, 64, 64),        # 131,072 voxels; needs a >=32-row position table
}


def make_graph(rng: np.random.Generator, stories: int, gx: int, gy: int,
               programs_per_story: Optional[Sequence[int]] = None,
               void_prob: float = 0.15, data_id: int = 0,
               voxel_size: Tuple[float, float, float] = (4.0, 3.0, 3.0)) -> VolumeDesignGraph:
    """One building with ``stories`` x ``gy`` x ``gx`` voxels ordered story-major."""
    F = int(stories)
    if programs_per_story is None:
        programs_per_story = [4] * F
    assert len(programs_per_story) == F and min(programs_per_story) >= 1

    # ---------------- program graph ----------------
    p_floor, p_class = [], []
    for f in range(F):
        n = int(programs_per_story[f])
        cls = rng.permutation(NUMBER_OF_CLASS)[: min(n, NUMBER_OF_CLASS)].tolist()
        while len(cls) < n:
            cls.append(int(rng.integers(NUMBER_OF_CLASS)))
        for c in cls:
            p_floor.append(f)
            p_class.append(int(c))
    Np = len(p_floor)
    p_floor_np = np.asarray(p_floor)
    first_of_story = np.searchsorted(p_floor_np, np.arange(F))
    adj = np.zeros((Np, Np), dtype=bool)
    for f in range(F):
        ids = np.nonzero(p_floor_np == f)[0]
        for a in range(len(ids) - 1):          # chain inside the story
            adj[ids[a], ids[a + 1]] = adj[ids[a + 1], ids[a]] = True
        if len(ids) > 2 and rng.random() < 0.5:  # close the ring sometimes
            adj[ids[0], ids[-1]] = adj[ids[-1], ids[0]] = True
        if f + 1 < F:                            # vertical circulation to the next story
            nxt = np.nonzero(p_floor_np == f + 1)[0]
            a, b = int(rng.choice(ids)), int(rng.choice(nxt))
            adj[a, b] = adj[b, a] = True
    src, dst = np.nonzero(adj)                   # source-major listing, bidirectional
    program_edge = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    same, diff = [[], []], [[], []]
    for i in range(Np):
        for j in range(i + 1, Np):
            if adj[i, j]:
                continue
            tgt = same if p_floor[i] == p_floor[j] else diff
            tgt[0] += [i, j]
            tgt[1] += [j, i]
    ratio = rng.random(NUMBER_OF_CLASS) + 0.05
    present = np.zeros(NUMBER_OF_CLASS, dtype=bool)
    present[np.asarray(p_class)] = True
    ratio = np.where(present, ratio, 0.0)
    ratio = ratio / ratio.sum()

    # ---------------- voxel graph ----------------
    Nv = F * gy * gx
    vid = np.arange(Nv).reshape(F, gy, gx)
    floor_idx = np.repeat(np.arange(F), gy * gx)
    yy = np.tile(np.repeat(np.arange(gy), gx), F)
    xx = np.tile(np.arange(gx), F * gy)
    dims = np.tile(np.asarray(voxel_size, dtype=np.float64), (Nv, 1))
    coord = np.stack([xx * voxel_size[0], yy * voxel_size[1], floor_idx * voxel_size[2]], axis=1)
    weight = rng.random(Nv) * 0.5 + 0.5
    voxel_feature = np.concatenate([dims / DIMENSION_NORM_FACTOR, coord / DIMENSION_NORM_FACTOR,
                                    weight[:, None]], axis=1).astype(np.float32)
    # neighbours listed per source voxel: -z, -y, -x, +x, +y, +z
    nb = np.full((Nv, 6), -1, dtype=np.int64)
    nb[vid[1:].ravel(), 0] = vid[:-1].ravel()
    nb[vid[:, 1:].ravel(), 1] = vid[:, :-1].ravel()
    nb[vid[:, :, 1:].ravel(), 2] = vid[:, :, :-1].ravel()
    nb[vid[:, :, :-1].ravel(), 3] = vid[:, :, 1:].ravel()
    nb[vid[:, :-1].ravel(), 4] = vid[:, 1:].ravel()
    nb[vid[:-1].ravel(), 5] = vid[1:].ravel()
    e_src = np.repeat(np.arange(Nv), 6)[nb.ravel() >= 0]
    e_dst = nb.ravel()[nb.ravel() >= 0]
    voxel_edge = torch.from_numpy(np.stack([e_src, e_dst]).astype(np.int64))
    proj = (yy * gx + xx).astype(np.int64)

    # labels: every voxel points at one program node of its story, or is void
    n_in_story = np.asarray(programs_per_story)[floor_idx]
    pick = first_of_story[floor_idx] + (rng.random(Nv) * n_in_story).astype(np.int64)
    void = rng.random(Nv) < void_prob
    label_node = np.where(void, -1, pick).astype(np.int64)
    voxel_label = np.zeros((Nv, NUMBER_OF_CLASS), dtype=np.float32)
    keep = np.nonzero(~void)[0]
    voxel_label[keep, np.asarray(p_class)[pick[keep]]] = 1.0

    # cross edges: every voxel of a story sees every program node of that story, voxel-major
    nvs = gy * gx
    cep, cev = [], []
    for f in range(F):
        ids = np.nonzero(p_floor_np == f)[0]
        cep.append(np.tile(ids, nvs))
        cev.append(np.repeat(np.arange(f * nvs, (f + 1) * nvs), len(ids)))
    cep = np.concatenate(cep).astype(np.int64)
    cev = np.concatenate(cev).astype(np.int64)

    edge_mask = (floor_idx[e_src] == floor_idx[e_dst]).astype(np.float32)
    onehot = np.eye(NUMBER_OF_CLASS, dtype=np.float32)[np.asarray(p_class)]
    return VolumeDesignGraph(
        data_id_str=str(data_id).zfill(6),
        program_floor_cluster=torch.tensor(p_floor, dtype=torch.long),
        story_level_feature=torch.tensor(p_floor, dtype=torch.float32),
        program_class_feature=torch.from_numpy(onehot),
        program_edge=program_edge,
        neg_same_story_program_edge=torch.tensor(same, dtype=torch.long).view(2, -1),
        neg_diff_story_program_edge=torch.tensor(diff, dtype=torch.long).view(2, -1),
        program_class_cluster=torch.tensor(p_class, dtype=torch.long),
        program_target_ratio=torch.from_numpy(ratio.astype(np.float32)),
        FAR=torch.tensor([float(rng.random() * 4 + 1)], dtype=torch.float32),
        voxel_floor_cluster=torch.from_numpy(floor_idx.astype(np.int64)),
        voxel_projection_cluster=torch.from_numpy(proj),
        voxel_feature=torch.from_numpy(voxel_feature),
        voxel_bool=torch.from_numpy((~void).astype(np.float32)),
        voxel_label=torch.from_numpy(voxel_label),
        voxel_edge=voxel_edge,
        voxel_edge_mask=torch.from_numpy(edge_mask),
        cross_edge_program_index_select=torch.from_numpy(cep),
        cross_edge_voxel_index_select=torch.from_numpy(cev),
        voxel_label_program_node_index=torch.from_numpy(label_node),
    )


def make_batch(preset: str = "dataset", seed: int = 0, graphs: Optional[int] = None,
               ragged: bool = False, follow_batch=DEFAULT_FOLLOW_BATCH) -> Batch:
    """A batch of synthetic buildings.  ``ragged`` varies stories / grid / programs per graph: True draws
    them anywhere up to the preset's size; "jitter" keeps every graph near the preset's size (stories F-2..F+1,
    grid +-2 cells, 2-6 program nodes per story), i.e. a stream of differently-shaped batches of one scale."""
    B, F, gx, gy = PRESETS[preset]
    if graphs is not None:
        B = graphs
    rng = np.random.default_rng(seed)
    data: List[VolumeDesignGraph] = []
    for g in range(B):
        if ragged == "jitter":
            f = int(rng.integers(max(1, F - 2), F + 2))
            x = int(rng.integers(max(1, gx - 2), gx + 3))
            y = int(rng.integers(max(1, gy - 2), gy + 3))
            pps = rng.integers(2, 7, size=f).tolist()
        elif ragged:
            f = int(rng.integers(1, F + 1))
            x = int(rng.integers(max(1, gx // 2), gx + 1))
            y = int(rng.integers(max(1, gy // 2), gy + 1))
            pps = rng.integers(1, 7, size=f).tolist()
        else:
            f, x, y, pps = F, gx, gy, [4] * F
        data.append(make_graph(rng, f, x, y, pps, data_id=g))
    return Batch.from_data_list(data, follow_batch)


def batch_sizes(batch) -> dict:
    return dict(B=int(batch.FAR.shape[0]), Np=int(batch.program_class_feature.shape[0]),
                Nv=int(batch.voxel_feature.shape[0]), Ep=int(batch.program_edge.shape[1]),
                Ev=int(batch.voxel_edge.shape[1]), Ec=int(batch.cross_edge_voxel_index_select.shape[0]),
                S=int(batch.voxel_floor_cluster.max().item()) + 1)
