"""Evaluation helpers with the reference's names (/root/reference/util_eval.py:73-157), device side.

``check_connectivity`` replaces the reference's three ``torch.unique(dim=1)`` sorts by a dense
program-by-program adjacency bitmap (program graphs are tiny: tens of nodes per building); edge
lists come back in the same lexicographic order ``unique`` produces.  JSON writing
(``save_output``) lives in ``json_io.py``.
"""
from __future__ import annotations

from typing import Optional

import torch

from .graph import Batch
from .util_graph import get_program_ratio


def check_connectivity(g, max_out_program_index, mask):
    """reference util_eval.py:102-136.  Returns (inter_edges, missing_edges, gen_edges), each [2, k]:
    program edges seen in the generated design, only in the input program graph, only generated."""
    ve = g.voxel_edge
    pe = g.program_edge
    Np = int(g.program_class_feature.shape[0])
    if ve.is_cuda:
        got = _check_connectivity_hip(ve, pe, Np, max_out_program_index, mask)
        if got is not None:
            return got
    on_edge = mask.reshape(-1)[ve]                       # [2, Ev]
    keep = on_edge.sum(dim=0) == 2                       # both end voxels unmasked (float compare, as the reference)
    masked_edges = ve[:, keep]
    if masked_edges.shape[1] == 0:                       # reference's else-branch
        return masked_edges, pe, masked_edges
    pred = max_out_program_index[masked_edges]           # [2, k] program node of either end
    flat = torch.zeros(Np * Np, dtype=torch.bool, device=ve.device)
    flat[pred[0] * Np + pred[1]] = True
    gt = torch.zeros(Np * Np, dtype=torch.bool, device=ve.device)
    gt[pe[0] * Np + pe[1]] = True

    def edges(bitmap):
        k = bitmap.nonzero(as_tuple=False).view(-1)      # ascending key = lexicographic (src, dst)
        return torch.stack((k // Np, k % Np))

    return edges(flat & gt), edges(gt & ~flat), edges(flat & ~gt)


def _check_connectivity_hip(ve, pe, Np, idx, mask):
    """bg_connectivity: the code table and the three sorted lists by three launches and ONE read-back (the counts).  None
    for inputs the entry point does not take (the bitmap form above then runs)."""
    from . import _lib
    m = mask.reshape(-1)
    Ev, Ep, Nv = int(ve.shape[1]), int(pe.shape[1]), int(m.shape[0])
    if (Ev == 0 or Np <= 0 or Np > 8192 or ve.dtype != torch.int64 or pe.dtype != torch.int64 or idx.dtype != torch.int64
            or m.dtype != torch.float32 or idx.numel() != Nv or not (ve.is_contiguous() and pe.is_contiguous()
                                                                  and idx.is_contiguous() and m.is_contiguous())):
        return None
    dev = ve.device
    cap_g = min(Ev, Np * Np)
    code = torch.empty(Np * Np, dtype=torch.int32, device=dev)
    inter, missing = torch.empty(2, max(Ep, 1), dtype=torch.int64, device=dev), torch.empty(2, max(Ep, 1), dtype=torch.int64, device=dev)
    gen = torch.empty(2, max(cap_g, 1), dtype=torch.int64, device=dev)
    counts = torch.empty(3, dtype=torch.int64, device=dev)
    _lib.check(_lib.load().bg_connectivity(ve.data_ptr(), Ev, m.data_ptr(), idx.data_ptr(), Nv, pe.data_ptr(), Ep, Np,
                                           code.data_ptr(), inter.data_ptr(), inter.shape[1], missing.data_ptr(),
                                           missing.shape[1], gen.data_ptr(), gen.shape[1], counts.data_ptr(), _lib.stream()),
               "bg_connectivity")
    ni, nm, ng = (int(x) for x in counts.tolist())          # the one synchronisation
    if ni + ng == 0:                                          # no voxel edge with both ends unmasked: the reference's else-branch
        empty = ve[:, :0]
        return empty, pe, empty
    return inter[:, :ni], missing[:, :nm], gen[:, :ng]


def draw_z(g, noise_dim: int, device, trunc: float = 1.0, generator=None):
    """z ~ U(0,1) drawn on the CPU generator and moved, as reference util_eval.py:81-87.  When ``generator`` is
    given and its voxel GNN takes no noise (reference models.py:239 builds it with noise_dim 0, so voxel_z is
    never read) voxel_z is not drawn: for a 10^6-voxel batch the unused draw and its copy cost more than the
    whole forward.  (The CPU generator's stream then differs from the reference's after this call.)"""
    program_z = torch.rand(g.program_class_feature.shape[0], noise_dim).to(device)
    if generator is not None and getattr(getattr(generator, "voxelGNN", None), "noise_dim", 1) == 0:
        voxel_z = None
    else:
        voxel_z = torch.rand(g.voxel_feature.shape[0], noise_dim).to(device)
    if trunc < 1.0:
        program_z.clamp_(min=-trunc, max=trunc)
        if voxel_z is not None:
            voxel_z.clamp_(min=-trunc, max=trunc)
    return program_z, voxel_z


def evaluate(data_loader, generator, raw_dir: Optional[str], output_dir: Optional[str], follow_batch, device_ids,
             number_of_batches: int = 0, trunc: float = 1.0, writer=None):
    """reference util_eval.py:73-99.  ``writer(batch, class_w, program_w, FAR, idx, out)`` is called per
    batch when given (json_io.save_output bound to raw_dir / output_dir); returns (acc, results)."""
    number_of_batches = min(number_of_batches, len(data_loader))
    device = device_ids[0] if device_ids else next(generator.parameters()).device
    results = []
    total_inter, total_program_edge = 0, 0
    from .plan import prefetch_plan
    it = iter(data_loader)

    def fetch(k):
        b = next(it, None) if k < number_of_batches else None
        if b is not None:
            b.to(device)
        return b
    with torch.no_grad():
        g, i = fetch(0), 0
        while g is not None:
            nxt = fetch(i + 1)
            if nxt is not None and torch.device(device).type == "cuda":
                prefetch_plan(nxt)                # the next design's index plan, beside this design's forward (side stream)
            program_z, voxel_z = draw_z(g, generator.noise_dim, device, trunc, generator)
            out, soft_out, mask, att, max_out_program_index = generator(g, program_z, voxel_z)
            inter_edges, missing_edges, gen_edges = check_connectivity(g, max_out_program_index, mask["hard"])
            total_inter += inter_edges.shape[1]
            total_program_edge += g.program_edge.shape[1]
            class_w, program_w, far = get_program_ratio(g, att["hard"], mask["hard"], area_index_in_voxel_feature=6)
            res = dict(out=out, max_out_program_index=max_out_program_index, class_weights=class_w,
                       program_weights=program_w, FAR=far, inter_edges=inter_edges, missing_edges=missing_edges,
                       gen_edges=gen_edges)
            if writer is not None:
                res["written"] = writer(g, class_w, program_w, far, max_out_program_index, out)
            results.append(res)
            g, i = nxt, i + 1
    acc = total_inter / max(total_program_edge, 1)
    print("acc=", acc)
    return acc, results


def generate_multiple_outputs_from_batch(batch, variation_num, generator, raw_dir, output_dir, follow_batch,
                                         device_ids, trunc: float = 1.0, writer=None):
    """reference util_eval.py:139-157: one forward over ``variation_num`` copies of a graph."""
    device = device_ids[0] if device_ids else next(generator.parameters()).device
    batch.to(device)
    with torch.no_grad():
        program_z, voxel_z = draw_z(batch, generator.noise_dim, device, trunc, generator)
        out, soft_out, mask, att, max_out_program_index = generator(batch, program_z, voxel_z)
        class_w, program_w, far = get_program_ratio(batch, att["hard"], mask["hard"], area_index_in_voxel_feature=6)
        if writer is not None:
            writer(batch, class_w, program_w, far, max_out_program_index, out, new_data_id_strs=list(range(variation_num)))
    return out, max_out_program_index, class_w, program_w, far


def generate_multiple_outputs_from_data(data, variation_num, generator, raw_dir, output_dir, follow_batch, device_ids):
    """reference util_eval.py:160-162."""
    batch = Batch.from_data_list([data for _ in range(variation_num)], follow_batch)
    return generate_multiple_outputs_from_batch(batch, variation_num, generator, raw_dir, output_dir, follow_batch,
                                                device_ids)
