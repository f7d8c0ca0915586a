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
    """First size[0] points of a U(low, high) box draw that fall inside the unit ball.  The draw is 3x oversized
    (acceptance is pi/6) and repeated whole in the improbable case it comes up short — the reference's scheme
    (utils/points.py:8-13), kept draw for draw so the torch CPU generator is consumed identically."""
    want, rest = size[0], tuple(size[1:])
    kept = ()
    while len(kept) < want:
        box = torch.zeros((3 * want,) + rest).uniform_(low, high)
        kept = box[box.norm(dim=1) < 1]
    return kept[:want]


def normalization_coef(config, epoch, normalize_points=None):
    """Radius c of the progressive hollow: linspace(0, 1, E)[epoch-1] up to epoch E, then 1; 0 when disabled."""
    rule = config['target_network_input']['normalization']
    enabled = rule['enable'] if normalize_points is None else normalize_points
    if not (enabled and rule['type'] == 'progressive'):
        return 0.0
    last = rule['epoch']
    return float(np.linspace(0, 1, last)[epoch - 1]) if epoch <= last else 1.0


def generate_points(config, epoch, size, normalize_points=None):
    """utils/points.py:16-36: ball points, those nearer the origin than c pushed out onto the sphere of radius c
    (norms through numpy on the fp32 values, as the reference takes them)."""
    c = normalization_coef(config, epoch, normalize_points)
    pts = generate_points_from_uniform_distribution(size=size)
    if c > 0.0:
        inner = np.linalg.norm(pts, axis=1) < c
        if inner.any():
            moved = pts[inner]
            radius = torch.from_numpy(np.linalg.norm(moved, axis=1)).float()
            pts[inner] = c * (moved.T / radius).T
    return pts


def sample_points_device(config, epoch, batch, n, device, seed, offset):
    return sample_points(batch, n, normalization_coef(config, epoch), seed, offset, device)
