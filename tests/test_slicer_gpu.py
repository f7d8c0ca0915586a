"""GPU: the on-device random-plane slicer (datasets/utils/dataset_generator.py:26-39).
The reference draws planes from numpy's global RNG, so parity is by properties: exact part sizes, a partition of the
input in original order, separation by the reported plane evaluated with the reference's formula, determinism per seed,
and the acceptance law (both orientations occur)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_slicer_partitions_exactly_and_in_order():
    from hyperpocket_amd.ops import slice_clouds
    g = torch.Generator(device="cuda").manual_seed(0)
    pts = torch.rand(48, 2048, 3, device="cuda", generator=g) - 0.5
    a, b, plane = slice_clouds(pts, 1024, seed=7)
    assert a.shape == (48, 1024, 3) and b.shape == (48, 1024, 3)
    P, A, Bn, pl = pts.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy(), plane.cpu().numpy()
    n_under_selected = 0
    for c in range(48):
        # reference formula: sign(dot(point, params) + bias) (HyperPlane.check_point)
        under = (P[c] @ pl[c, :3] + pl[c, 3]) > 0
        if under.sum() == 1024 and np.array_equal(P[c][under], A[c]):
            n_under_selected += 1
            assert np.array_equal(P[c][~under], Bn[c])
        else:
            assert (~under).sum() == 1024 and np.array_equal(P[c][~under], A[c]) and np.array_equal(P[c][under], Bn[c])
    assert 0 < n_under_selected <= 48
    a2, b2, _ = slice_clouds(pts, 1024, seed=7)
    assert torch.equal(a, a2) and torch.equal(b, b2)
    a3, _, _ = slice_clouds(pts, 1024, seed=8)
    assert not torch.equal(a, a3)


def test_slicer_uneven_target_and_dataset_generator_api():
    from hyperpocket_amd.datasets.utils.dataset_generator import SlicedDatasetGenerator
    g = torch.Generator(device="cuda").manual_seed(1)
    pts = torch.rand(333, 3, device="cuda", generator=g)
    part, rest = SlicedDatasetGenerator.generate_item(pts, 100, seed=3)
    assert part.shape == (100, 3) and rest.shape == (233, 3)
    allp = torch.cat([part, rest]).cpu().numpy()
    assert sorted(map(tuple, allp.tolist())) == sorted(map(tuple, pts.cpu().numpy().tolist()))
    batch = torch.rand(5, 2048, 3, device="cuda", generator=g) - 0.5
    ex, mi = SlicedDatasetGenerator.generate_batch(batch, 1024, seed=11)
    assert ex.shape == (5, 1024, 3) and mi.shape == (5, 1024, 3)
