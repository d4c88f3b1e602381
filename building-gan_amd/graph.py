"""Batch data layout of the hot path: a PyG-free restatement of the reference's
``VolumeDesignGraph`` container and of ``Batch.from_data_list``.

Reference contract:
  * attribute list and meaning ........ /root/reference/Data/VolumeDesignGraph.py:5-51
  * per-key index increments .......... /root/reference/Data/VolumeDesignGraph.py:60-72
  * per-key concatenation dimension ... /root/reference/Data/VolumeDesignGraph.py:74-77
  * ``follow_batch`` vectors .......... /root/reference/train.py:43,50

Only host (CPU) tensor plumbing lives here; no kernels.  The models read attributes
off any object that exposes them, so a real PyG ``Batch`` works as well.
"""
from __future__ import annotations

import copy
from typing import Dict, Iterable, List, Optional, Sequence

import torch

# increments: which node count an index attribute is offset by when graphs are concatenated
_PROGRAM_INC = ("program_edge", "cross_edge_program_index_select", "voxel_label_program_node_index",
                "neg_same_story_program_edge", "neg_diff_story_program_edge")
_VOXEL_INC = ("voxel_edge", "cross_edge_voxel_index_select")
_FLOOR_INC = ("program_floor_cluster", "voxel_floor_cluster")
_CLASS_INC = ("program_class_cluster",)
_PROJ_INC = ("voxel_projection_cluster",)
_LAST_DIM = ("program_edge", "voxel_edge", "neg_same_story_program_edge", "neg_diff_story_program_edge")

GRAPH_KEYS = (
    "data_id_str", "program_floor_cluster", "story_level_feature", "program_class_feature",
    "program_edge", "neg_same_story_program_edge", "neg_diff_story_program_edge",
    "program_class_cluster", "program_target_ratio", "FAR",
    "voxel_floor_cluster", "voxel_projection_cluster", "voxel_feature", "voxel_bool", "voxel_label",
    "voxel_edge", "voxel_edge_mask", "cross_edge_program_index_select",
    "cross_edge_voxel_index_select", "voxel_label_program_node_index",
)

DEFAULT_FOLLOW_BATCH = ("program_target_ratio", "program_class_feature", "voxel_feature")


class VolumeDesignGraph:
    """One building: a program graph, a voxel graph and the cross edges between them."""

    def __init__(self, **kwargs):
        for k in GRAPH_KEYS:
            setattr(self, k, kwargs.pop(k, None))
        for k, v in kwargs.items():
            setattr(self, k, v)

    # ---- mapping-style access, as PyG Data offers -----------------------------------------
    @property
    def keys(self) -> List[str]:
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("_")]

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return getattr(self, key, None) is not None

    # ---- batching rules ---------------------------------------------------------------------
    def __inc__(self, key, value=None) -> int:
        if key in _PROGRAM_INC:
            return int(self.program_class_feature.size(0))
        if key in _VOXEL_INC:
            return int(self.voxel_feature.size(0))
        if key in _FLOOR_INC:
            return int(self.story_level_feature.max().item() + 1)
        if key in _CLASS_INC:
            return int(self.program_class_feature.size(1))
        if key in _PROJ_INC:
            return int(self.voxel_projection_cluster.max().item() + 1)
        return 0

    def __cat_dim__(self, key, value=None) -> int:
        return -1 if key in _LAST_DIM else 0

    def to(self, device, non_blocking: bool = False):
        """As ``batch.to(device)`` of reference train.py:75.  An event is recorded behind the last copy
        (``_bg_h2d_event``; with ``non_blocking=True`` -- copies from pinned memory -- the host has not waited for
        them): streams other than the current one -- the plan's prefetch stream -- order themselves behind THAT event
        instead of behind everything queued on the current stream, so the next batch's plan runs beside the iteration
        in flight rather than after it.  Code that rewrites a batch's tensors with device ops afterwards must drop the
        attribute (or move the batch again)."""
        moved = False
        for k in self.keys:
            v = getattr(self, k)
            if torch.is_tensor(v):
                w = v.to(device, non_blocking=non_blocking)
                if w is not v:
                    moved = True
                    setattr(self, k, w)
        if not moved:
            # every tensor is where it was asked to be (util_eval's ``batch.to(device)`` on a resident batch): the same
            # tensors, so the index plan built from them -- and a plan in flight -- stay valid
            return self
        self.__dict__.pop("_bg_plan", None)  # device-side plans do not survive a move
        self.__dict__.pop("_bg_plan_pending", None)
        self.__dict__.pop("_bg_h2d_event", None)
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        if dev.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self.__dict__["_bg_h2d_event"] = ev
        return self

    def pin_memory(self):
        """Page-locked host copies of every tensor (what a DataLoader with ``pin_memory=True`` hands over)."""
        for k in self.keys:
            v = getattr(self, k)
            if torch.is_tensor(v) and not v.is_cuda:
                setattr(self, k, v.pin_memory())
        return self


class Batch(VolumeDesignGraph):
    """Several ``VolumeDesignGraph`` concatenated into one disconnected graph."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.__slices__: Optional[Dict[str, List[int]]] = None
        self.__data_class__ = VolumeDesignGraph
        self.num_graphs: int = 0
        self.num_stories: Optional[int] = None      # total stories of the batch, known at collate time (the plan needs it)

    @property
    def keys(self) -> List[str]:
        return [k for k, v in self.__dict__.items()
                if v is not None and not k.startswith("_") and k not in ("num_graphs", "num_stories")]

    @staticmethod
    def from_data_list(data_list: Sequence[VolumeDesignGraph],
                       follow_batch: Iterable[str] = DEFAULT_FOLLOW_BATCH) -> "Batch":
        follow_batch = DEFAULT_FOLLOW_BATCH if follow_batch is None else tuple(follow_batch)
        batch = Batch()
        batch.num_graphs = len(data_list)
        keys = [k for k in GRAPH_KEYS if data_list[0][k] is not None]
        slices: Dict[str, List[int]] = {k: [0] for k in keys}
        parts: Dict[str, list] = {k: [] for k in keys}
        follow: Dict[str, list] = {k: [] for k in follow_batch}
        cumsum = {k: 0 for k in keys}
        for gi, data in enumerate(data_list):
            for k in keys:
                item = data[k]
                if torch.is_tensor(item):
                    if item.dtype != torch.bool and cumsum[k] != 0:
                        item = item + cumsum[k]
                    dim = data.__cat_dim__(k, item)
                    n = item.size(dim) if item.dim() > 0 else 1
                    if item.dim() == 0:
                        item = item.view(1)
                    parts[k].append(item)
                    slices[k].append(slices[k][-1] + n)
                    if k in follow:
                        follow[k].append(torch.full((n,), gi, dtype=torch.long))
                else:
                    parts[k].append(item)
                    slices[k].append(slices[k][-1] + 1)
                cumsum[k] += data.__inc__(k, data[k])
        for k in keys:
            if torch.is_tensor(parts[k][0]):
                batch[k] = torch.cat(parts[k], dim=data_list[0].__cat_dim__(k, parts[k][0]))
            else:
                batch[k] = list(parts[k])
        for k in follow_batch:
            batch[k + "_batch"] = torch.cat(follow[k], dim=0)
        if "voxel_floor_cluster" in cumsum:
            # the running story offset after the last graph = the number of (dense, global) story ids: the plan
            # otherwise reads voxel_floor_cluster.max() back from the device (one more synchronisation per batch)
            batch.num_stories = int(cumsum["voxel_floor_cluster"])
        batch.__slices__ = slices
        return batch


def detach_batch(batch):
    """Re-wrap a batch with detached tensors (/root/reference/util_graph.py:8-18).

    The device-side segment plan (``_bg_plan``) carries no gradient, so it is shared.
    """
    out = Batch()
    out.__slices__ = copy.deepcopy(getattr(batch, "__slices__", None))
    out.num_graphs = getattr(batch, "num_graphs", 0)
    out.num_stories = getattr(batch, "num_stories", None)
    for k in batch.keys:
        v = batch[k]
        out[k] = v.detach() if torch.is_tensor(v) else copy.deepcopy(v)
    plan = batch.__dict__.get("_bg_plan") if hasattr(batch, "__dict__") else None
    if plan is not None:
        out.__dict__["_bg_plan"] = plan
    return out
