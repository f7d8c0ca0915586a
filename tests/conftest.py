"""Shared test plumbing: marker registration, import paths, oracle loaders.

The oracle (oracle/) is test infrastructure: it is imported only from here, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg — never by the product
package (3d-point-clouds-autocomplete_amd/).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

os.environ.setdefault("OMP_NUM_THREADS", "16")   # the C oracle parallelises over clouds; a 256-core GPU host oversubscribes badly

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "3d-point-clouds-autocomplete_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def fixture_state_(tensors, g):
    """Bring seeded-init parameters to the state a model fixture was captured at.  `tensors`: name -> tensor under the
    reference's state_dict names (a module's state_dict() or the oracle's parameter dict), modified in place.  Plain
    fixtures are captured at the seeded init: nothing to do.  model_trained.npz (tests/golden/make_golden.py,
    train_to_operating_point) carries a recipe: the hypernetwork heads' weights times 2**head_scale_log2 (exact), then
    the trained values of the small tensors the reference's train_epoch updated."""
    if "head_scale_log2" not in g:
        return tensors
    import torch
    scale = 2.0 ** int(g["head_scale_log2"])
    with torch.no_grad():
        for k, t in tensors.items():
            if k.startswith("hyper_network.output.") and k.endswith(".weight"):
                t.mul_(scale)
            key = "trained__" + k.replace(".", "__")
            if key in g:
                t.copy_(torch.from_numpy(g[key]).to(t.dtype))
    return tensors


class OracleLib:
    """ctypes view of oracle/libstructural_losses_ref.so (the C restatement)."""

    def __init__(self):
        so = os.path.join(ROOT, "oracle", "libstructural_losses_ref.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        self.lib = ctypes.CDLL(so)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def nndistance(self, xyz1, xyz2):
        xyz1, xyz2 = np.ascontiguousarray(xyz1, np.float32), np.ascontiguousarray(xyz2, np.float32)
        b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        d1, i1 = np.empty((b, n), np.float32), np.empty((b, n), np.int32)
        d2, i2 = np.empty((b, m), np.float32), np.empty((b, m), np.int32)
        self.lib.ref_nndistance(b, n, self._p(xyz1), m, self._p(xyz2), self._p(d1), self._p(i1), self._p(d2), self._p(i2))
        return d1, i1, d2, i2

    def nndistancegrad(self, xyz1, xyz2, gd1, i1, gd2, i2):
        xyz1, xyz2 = np.ascontiguousarray(xyz1, np.float32), np.ascontiguousarray(xyz2, np.float32)
        gd1, gd2 = np.ascontiguousarray(gd1, np.float32), np.ascontiguousarray(gd2, np.float32)
        i1, i2 = np.ascontiguousarray(i1, np.int32), np.ascontiguousarray(i2, np.int32)
        b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
        self.lib.ref_nndistancegrad(b, n, self._p(xyz1), m, self._p(xyz2), self._p(gd1), self._p(i1),
                                    self._p(gd2), self._p(i2), self._p(g1), self._p(g2))
        return g1, g2

    # which fma contractions of the reference source the HIP kernels implement (oracle/structural_losses_ref.c): nvcc's
    # default -fmad=true applied to the phase sums and phase 3 — bits 0 and 1
    KERNEL_CONTRACT = 3

    def approxmatch(self, xyz1, xyz2, contract=KERNEL_CONTRACT):
        """contract: which fma contractions of the reference source to assume (oracle/structural_losses_ref.c);
        0 = the literal source (every product rounded before its add), 3 = nvcc's default contraction (the kernels')."""
        xyz1, xyz2 = np.ascontiguousarray(xyz1, np.float32), np.ascontiguousarray(xyz2, np.float32)
        b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        match = np.empty((b, m, n), np.float32)
        temp = np.empty((b, 2 * (n + m)), np.float32)
        self.lib.ref_approxmatch_ex(b, n, m, self._p(xyz1), self._p(xyz2), self._p(match), self._p(temp), int(contract))
        return match, temp

    def approxmatch_f64(self, xyz1, xyz2):
        """The nine-level algorithm in fp64 with libm exp: (match (b,m,n) float64, cost (b,) float64)."""
        xyz1, xyz2 = np.ascontiguousarray(xyz1, np.float32), np.ascontiguousarray(xyz2, np.float32)
        b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        match = np.empty((b, m, n), np.float64)
        cost = np.empty((b,), np.float64)
        self.lib.ref_approxmatch_f64(b, n, m, self._p(xyz1), self._p(xyz2), self._p(match), self._p(cost))
        return match, cost

    def matchcost(self, xyz1, xyz2, match):
        xyz1, xyz2 = np.ascontiguousarray(xyz1, np.float32), np.ascontiguousarray(xyz2, np.float32)
        match = np.ascontiguousarray(match, np.float32)
        b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        out = np.empty((b,), np.float32)
        self.lib.ref_matchcost(b, n, m, self._p(xyz1), self._p(xyz2), self._p(match), self._p(out))
        return out

    def matchcostgrad(self, xyz1, xyz2, match):
        xyz1, xyz2 = np.ascontiguousarray(xyz1, np.float32), np.ascontiguousarray(xyz2, np.float32)
        match = np.ascontiguousarray(match, np.float32)
        b, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
        self.lib.ref_matchcostgrad(b, n, m, self._p(xyz1), self._p(xyz2), self._p(match), self._p(g1), self._p(g2))
        return g1, g2


@pytest.fixture(scope="session")
def oracle_lib():
    return OracleLib()


@pytest.fixture(scope="session")
def ref():
    """The torch-CPU restatement of the model step (oracle/hyperpocket_ref.py)."""
    from oracle import hyperpocket_ref
    return hyperpocket_ref
