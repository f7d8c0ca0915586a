"""Decoder input points (utils/points.py:8-36).

Two producers of the same distribution (uniform in the unit ball, points with |p| < c pushed out
to radius c, c = linspace(0,1,E)[epoch-1]):
  * ``generate_points`` / ``generate_points_from_uniform_distribution`` — the reference's host
    routine, draw for draw (torch global CPU generator), for callers that need its exact values;
  * ``sample_points_device`` — all B clouds of a step in one HIP launch (Philox counter RNG), what
    FullModel uses by default: the reference's per-cloud CPU draw + H2D copy (model/full_model.py:
    72-74, ~1 ms per cloud) would otherwise dominate a ~ms GPU step.
"""
import numpy as np
import torch

from ..ops import sample_points


def generate_points_from_uniform_distribution(size, low=-1, high=1):
    while True:
        points = torch.zeros([size[0] * 3, *size[1:]]).uniform_(low, high)
        points = points[torch.norm(points, dim=1) < 1]
        if points.shape[0] >= size[0]:
            return points[:size[0]]


def normalization_coef(config, epoch, normalize_points=None):
    norm = config['target_network_input']['normalization']
    if normalize_points is None:
        normalize_points = norm['enable']
    if normalize_points and norm['type'] == 'progressive':
        max_epoch = norm['epoch']
        return float(np.linspace(0, 1, max_epoch)[epoch - 1]) if epoch <= max_epoch else 1.0
    return 0.0


def generate_points(config, epoch, size, normalize_points=None):
    coef = normalization_coef(config, epoch, normalize_points)
    points = generate_points_from_uniform_distribution(size=size)
    if coef > 0.0:
        norms = np.linalg.norm(points, axis=1)
        sel = norms < coef
        if sel.any():
            sub = points[sel]
            points[sel] = coef * (sub.T / torch.from_numpy(np.linalg.norm(sub, axis=1)).float()).T
    return points


def sample_points_device(config, epoch, batch, n, device, seed, offset):
    return sample_points(batch, n, normalization_coef(config, epoch), seed, offset, device)
