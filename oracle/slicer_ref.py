"""TEST INFRASTRUCTURE (the checker, never the product): numpy restatement of the reference's random-plane slicer with the
candidate planes as an input.  Pinned by tests/golden/slicer.npz (outputs of the reference's own
datasets/utils/dataset_generator.py under a seeded np.random; tests/golden/make_golden_slicer.py).

check_point      -> /root/reference/datasets/utils/dataset_generator.py:10-11   sign(dot(point, params) + bias), float64
generate_item    -> :29-39   first candidate whose "under" side (check > 0) or, failing that, whose other side has exactly
                             `target` points; returns (that side, the rest), both in the cloud's point order
"""
import numpy as np


def slice_with_planes(points, planes, target=1024):
    """points (N,3) float32, planes (R,4) float64 rows (params, bias) -> (part_a, part_b, index of the accepted plane);
    (None, None, -1) if none of the R candidates is accepted."""
    points = np.asarray(points)
    planes = np.asarray(planes, np.float64)
    for r in range(planes.shape[0]):
        under = np.sign(np.dot(points, planes[r, :3]) + planes[r, 3]) > 0
        if target == int(under.sum()):
            return points[under], points[~under], r
        if target == int((~under).sum()):
            return points[~under], points[under], r
    return None, None, -1
