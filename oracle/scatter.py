"""torch_scatter / torch_geometric primitives restated with stock CPU PyTorch (oracle only).

Upstream semantics followed:
  * scatter(..., reduce='sum') == scatter_add: sequential accumulation in index order on CPU.
  * scatter(..., reduce='mean'): sum / clamp(count, min=1)  (an empty row stays 0).
  * scatter_max: empty row -> value 0, arg = src.size(dim); ties keep the FIRST index
    (CPU rule: a later element replaces the running max only when strictly greater);
    the gradient goes to the arg element only.
  * avg_pool_x / max_pool_x with ``size``: dim_size = (batch.max()+1)*size, second result None;
    without ``size``: clusters relabelled by sorted unique, batch_out = batch[perm].
  * global_max_pool(x, batch): scatter-max into batch.max()+1 rows.
Call sites in the reference: /root/reference/Model/models.py:24,71,123,134,138,166-167,262-263,
/root/reference/util.py:20-24, /root/reference/util_graph.py:32-46.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def maybe_num_nodes(index: torch.Tensor, num_nodes: Optional[int] = None) -> int:
    return int(index.max()) + 1 if num_nodes is None else num_nodes


def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add(0, index, src)


scatter_add = scatter_sum


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    total = scatter_sum(src, index, dim_size)
    count = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
    count = count.index_add(0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
    if src.is_floating_point():
        return total / count.view((-1,) + (1,) * (src.dim() - 1))
    return torch.div(total, count.view((-1,) + (1,) * (src.dim() - 1)), rounding_mode="floor")


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (values, arg) reduced over dim 0; src is [E] or [E, C]."""
    E = src.size(0)
    flat = src.reshape(E, -1)
    C = flat.size(1)
    with torch.no_grad():
        idx2 = index.view(-1, 1).expand(E, C)
        if flat.is_floating_point():
            lowest = torch.finfo(flat.dtype).min
        else:
            lowest = torch.iinfo(flat.dtype).min
        best = torch.full((dim_size, C), lowest, dtype=flat.dtype, device=flat.device)
        best = best.scatter_reduce(0, idx2, flat, reduce="amax", include_self=True)
        pos = torch.arange(E, device=flat.device).view(-1, 1).expand(E, C)
        cand = torch.where(flat == best.gather(0, idx2), pos, torch.full_like(pos, E))
        arg = torch.full((dim_size, C), E, dtype=torch.long, device=flat.device)
        arg = arg.scatter_reduce(0, idx2, cand, reduce="amin", include_self=True)
        empty = arg == E
    safe = arg.clamp(max=max(E - 1, 0))
    vals = flat.gather(0, safe) if E > 0 else torch.zeros((dim_size, C), dtype=flat.dtype)
    vals = torch.where(empty, torch.zeros_like(vals), vals)
    out_shape = (dim_size,) + tuple(src.shape[1:])
    return vals.reshape(out_shape), arg.reshape(out_shape)


def scatter(src, index, dim_size, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim_size)
    if reduce == "max":
        return scatter_max(src, index, dim_size)[0]
    raise ValueError(reduce)


def consecutive_cluster(src: torch.Tensor):
    unique, inv = torch.unique(src, sorted=True, return_inverse=True)
    perm = torch.arange(inv.size(0), dtype=inv.dtype, device=inv.device)
    perm = inv.new_empty(unique.size(0)).scatter_(0, inv, perm)
    return inv, perm


def avg_pool_x(cluster, x, batch, size: Optional[int] = None):
    if size is not None:
        batch_size = int(batch.max().item()) + 1
        return scatter_mean(x, cluster, batch_size * size), None
    cluster, perm = consecutive_cluster(cluster)
    x = scatter_mean(x, cluster, int(cluster.max()) + 1)
    return x, batch[perm]


def max_pool_x(cluster, x, batch, size: Optional[int] = None):
    if size is not None:
        batch_size = int(batch.max().item()) + 1
        return scatter_max(x, cluster, batch_size * size)[0], None
    cluster, perm = consecutive_cluster(cluster)
    x = scatter_max(x, cluster, int(cluster.max()) + 1)[0]
    return x, batch[perm]


def global_max_pool(x, batch):
    return scatter_max(x, batch, int(batch.max().item()) + 1)[0]


def propagate_mean(message_fn, edge_index: torch.Tensor, num_nodes: int, **node_feats):
    """MessagePassing.propagate, default flow source_to_target, aggr='mean'.

    ``*_j`` = feature[edge_index[0]] (source), ``*_i`` = feature[edge_index[1]] (target);
    messages are mean-aggregated onto the target node.
    """
    src, dst = edge_index[0], edge_index[1]
    lifted = {}
    for name, feat in node_feats.items():
        lifted[name + "_j"] = feat.index_select(0, src)
        lifted[name + "_i"] = feat.index_select(0, dst)
    msg = message_fn(**lifted)
    return scatter_mean(msg, dst, num_nodes)


def glorot_(tensor: torch.Tensor) -> torch.Tensor:
    a = (6.0 / (tensor.size(-2) + tensor.size(-1))) ** 0.5
    with torch.no_grad():
        tensor.uniform_(-a, a)
    return tensor
