"""Oracle for the integer segment plans (TEST INFRASTRUCTURE ONLY; numpy, bit-exact bar).

The reference never builds CSR structures itself: it hands COO index vectors to torch_scatter,
whose CPU kernels accumulate in index order.  A plan that keeps
that order is therefore a *stable* counting sort of the edges by segment id; these functions
state that with numpy so the device-built plans can be compared entry for entry.
"""
from __future__ import annotations

import numpy as np


def csr_by_key(key: np.ndarray, other: np.ndarray, num_rows: int):
    """Stable sort of edges by ``key``; returns (rowptr[int32 n+1], col[int32 E], perm[int32 E]).

    ``col[k] = other[perm[k]]``; inside a row, edges keep their original (COO) order.
    """
    key = np.asarray(key, dtype=np.int64)
    perm = np.argsort(key, kind="stable")
    counts = np.bincount(key, minlength=num_rows)
    rowptr = np.zeros(num_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    col = np.asarray(other, dtype=np.int64)[perm]
    return rowptr.astype(np.int32), col.astype(np.int32), perm.astype(np.int32)


def segment_ptr(seg_id: np.ndarray, num_segments: int) -> np.ndarray:
    """Pointer array of a non-decreasing segment-id vector (lower bounds)."""
    seg_id = np.asarray(seg_id, dtype=np.int64)
    assert np.all(seg_id[1:] >= seg_id[:-1]), "segment ids must be non-decreasing"
    return np.searchsorted(seg_id, np.arange(num_segments + 1), side="left").astype(np.int32)


def floor_level(vfc: np.ndarray, vbi: np.ndarray) -> np.ndarray:
    """reference models.py:22-25: story id minus the smallest story id of the voxel's graph."""
    vfc = np.asarray(vfc, dtype=np.int64)
    vbi = np.asarray(vbi, dtype=np.int64)
    first = np.full(int(vbi.max()) + 1, np.iinfo(np.int64).max)
    np.minimum.at(first, vbi, vfc)
    return (vfc - first[vbi]).astype(np.int32)


def inv_degree(rowptr: np.ndarray) -> np.ndarray:
    deg = np.diff(rowptr.astype(np.int64))
    return (np.float32(1.0) / np.maximum(deg, 1).astype(np.float32)).astype(np.float32)


def chunk_table(segptr: np.ndarray, rows_per_chunk: int):
    """Chunks of at most ``rows_per_chunk`` rows that never straddle a segment boundary.

    Returns (chunk_seg[int32 C], chunk_start[int32 C], chunk_end[int32 C], seg_chunk_ptr[int32 S+1]).
    """
    segptr = np.asarray(segptr, dtype=np.int64)
    lens = np.diff(segptr)
    n_chunks = (lens + rows_per_chunk - 1) // rows_per_chunk
    cptr = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(n_chunks, out=cptr[1:])
    seg = np.repeat(np.arange(len(lens)), n_chunks)
    local = np.arange(int(cptr[-1])) - cptr[seg]
    start = segptr[seg] + local * rows_per_chunk
    end = np.minimum(start + rows_per_chunk, segptr[seg + 1])
    return seg.astype(np.int32), start.astype(np.int32), end.astype(np.int32), cptr.astype(np.int32)
