"""utils/evaluation/mmd.py:23-47 — minimum matching distance of a reference set against generated samples,
over the HIP nearest-neighbour kernel.

Faithful to the reference including its quirk: ``nn_distance(ref (1,N,3), chunk (<=batch,N,3))`` takes
the batch size from its FIRST argument, so only the first cloud of every chunk is ever compared.
"""
import numpy as np
import torch

from ..pytorch_structural_losses.nn_distance import nn_distance


def iterate_in_chunks(seq, n):
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def minimum_mathing_distance(sample_pcs, ref_pcs, batch_size, device=None):
    n_ref, n_pc_points, pc_dim = ref_pcs.shape
    _, n_pc_points_s, pc_dim_s = sample_pcs.shape
    if n_pc_points != n_pc_points_s or pc_dim != pc_dim_s:
        raise ValueError('Incompatible size of point-clouds.')
    matched_dists = []
    for i in range(n_ref):
        ref = torch.from_numpy(ref_pcs[i]).unsqueeze(0).to(device).contiguous()
        best = []
        for chunk_np in iterate_in_chunks(sample_pcs, batch_size):
            chunk = torch.from_numpy(chunk_np).to(device).contiguous()
            ref_to_s, s_to_ref = nn_distance(ref, chunk)       # b = 1: chunk[0] only (reference behaviour)
            best.append(torch.min(ref_to_s.mean(dim=1) + s_to_ref.mean(dim=1)).item())
        matched_dists.append(np.min(best))
    return np.mean(matched_dists), matched_dists
