"""Oracle restatement of /root/reference/util_eval.py:102-136 check_connectivity (TEST INFRASTRUCTURE)."""
import torch


def check_connectivity(g, max_out_program_index, mask):
    on_edge = mask.reshape([-1])[g.voxel_edge]
    sums = torch.sum(on_edge, dim=0)
    masked_edges = g.voxel_edge[:, sums == 2]
    if masked_edges.shape[1] != 0:
        predicted = torch.unique(max_out_program_index[masked_edges], dim=1)
        uniq, cnt = torch.cat((g.program_edge, predicted), dim=1).unique(return_counts=True, dim=1)
        inter = uniq[:, cnt > 1]
        uniq, cnt = torch.cat((g.program_edge, inter), dim=1).unique(return_counts=True, dim=1)
        missing = uniq[:, cnt == 1]
        uniq, cnt = torch.cat((predicted, inter), dim=1).unique(return_counts=True, dim=1)
        gen = uniq[:, cnt == 1]
    else:
        inter, missing, gen = masked_edges, g.program_edge, masked_edges
    return inter, missing, gen
