#!/usr/bin/env python3
"""Golden fixture for the random-plane slicer: tests/golden/slicer.npz.

Runs the REFERENCE's own datasets/utils/dataset_generator.py (imported by file path: it needs only numpy) under a seeded
np.random and records, per case, the input cloud, the candidate planes HyperPlane.get_random_plane handed the loop —
(params, bias) float64, up to the accepted one plus a few the reference never drew, so that "first accepted wins" is
tested too — the index of the accepted plane and both returned parts.  Nothing of the reference travels: the fixture is
inputs + expected outputs.  Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_slicer.py
"""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("HP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("ref_dataset_generator", os.path.join(REF, "datasets/utils/dataset_generator.py"))
dg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(dg)

_recorded = []
_orig = dg.HyperPlane.get_random_plane


def _recording_get_random_plane():
    pl = _orig()                                  # the reference's own draw (np.random.rand(3, 3)) and plane formula
    _recorded.append(np.concatenate([np.asarray(pl.params, np.float64), [np.float64(pl.bias)]]))
    return pl


dg.HyperPlane.get_random_plane = staticmethod(_recording_get_random_plane)

# (name, N, target, cloud maker).  The reference's clouds live in the +-0.5 cube (utils/util.py:88) while the plane's three
# points are uniform in [0,1)^3: most candidates miss the cloud entirely, an exact split takes hundreds to thousands of draws.
CASES = [
    ("epn_2048_1024", 2048, 1024, lambda r: (r.random((2048, 3)) - 0.5).astype(np.float32)),        # generate_item's default
    ("epn_2048_1024_b", 2048, 1024, lambda r: (r.standard_normal((2048, 3)) * 0.15).astype(np.float32)),
    ("small_256_128", 256, 128, lambda r: (r.random((256, 3)) - 0.5).astype(np.float32)),
    ("uneven_333_100", 333, 100, lambda r: r.random((333, 3)).astype(np.float32)),                     # either side may match
    ("uneven_500_77", 500, 77, lambda r: (r.random((500, 3)) * 0.8 + 0.1).astype(np.float32)),
    ("offset_300_150", 300, 150, lambda r: (r.random((300, 3)) - 0.25).astype(np.float32)),
]

out = {}
rs = np.random.RandomState(2020)                  # cloud coordinates: a private generator, not the global one the planes use
for ci, (name, N, target, make) in enumerate(CASES):
    pts = make(rs)
    np.random.seed(1000 + ci)
    del _recorded[:]
    a, b = dg.SlicedDatasetGenerator.generate_item(pts, target)
    accepted = len(_recorded) - 1
    for _ in range(9):                            # candidates past the accepted one: must never be looked at
        _recording_get_random_plane()
    planes = np.stack(_recorded).astype(np.float64)
    assert a.shape == (target, 3) and b.shape == (N - target, 3) and a.dtype == np.float32
    out[name + "_points"] = pts
    out[name + "_planes"] = planes
    out[name + "_accepted"] = np.int32(accepted)
    out[name + "_part_a"] = np.ascontiguousarray(a)
    out[name + "_part_b"] = np.ascontiguousarray(b)
    print(f"{name}: N={N} target={target} accepted candidate {accepted} of {len(planes)}")
out["cases"] = np.array([c[0] for c in CASES])
np.savez_compressed(os.path.join(OUT, "slicer.npz"), **out)
print("wrote", os.path.join(OUT, "slicer.npz"), os.path.getsize(os.path.join(OUT, "slicer.npz")), "bytes")
