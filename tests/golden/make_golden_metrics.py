#!/usr/bin/env python3
"""Golden fixtures for the evaluation consumers of the structural losses:
utils/metrics.py (emd_approx, EMD_CD, _pairwise_EMD_CD_, mmd_cov, compute_all_metrics, knn) and
utils/evaluation/mmd.py (minimum_mathing_distance), produced by running the REFERENCE's own host code on CPU.

The reference's compiled module `StructuralLossesBackend` cannot be built here (nvcc/ATen-CUDA), so its five
functions are supplied by the CPU oracle (oracle/libstructural_losses_ref.so) through a stand-in module; everything
above that line — batching, expansion of the sample cloud, means, mins, coverage — is the reference's code.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
from conftest import OracleLib  # noqa: E402

REF = os.environ.get("HP_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))
lib = OracleLib()


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if dtype is None else t.to(dtype)


backend = types.ModuleType("utils.pytorch_structural_losses.StructuralLossesBackend")
backend.ApproxMatch = lambda a, b: [_t(x) for x in lib.approxmatch(a.numpy(), b.numpy())]
backend.MatchCost = lambda a, b, m: _t(lib.matchcost(a.numpy(), b.numpy(), m.numpy()))
backend.MatchCostGrad = lambda a, b, m: [_t(x) for x in lib.matchcostgrad(a.numpy(), b.numpy(), m.numpy())]
# b and n come from the first argument (structural_loss.cpp:86-93) — the oracle wrapper does the same
backend.NNDistance = lambda a, b: [_t(x) for x in lib.nndistance(a.numpy(), b.numpy()[:a.shape[0]])]
backend.NNDistanceGrad = lambda a, b, i1, i2, g1, g2: [_t(x) for x in lib.nndistancegrad(
    a.numpy(), b.numpy(), g1.numpy(), i1.numpy(), g2.numpy(), i2.numpy())]
sys.modules[backend.__name__] = backend

import utils.metrics as ref_metrics  # noqa: E402
import utils.evaluation.mmd as ref_mmd  # noqa: E402
from losses.champfer_loss import ChamferLoss  # noqa: E402


def main():
    g = torch.Generator().manual_seed(41)
    sample = torch.rand(5, 96, 3, generator=g) - 0.5
    ref = torch.rand(7, 96, 3, generator=g) - 0.5
    out = {"sample": sample.numpy(), "ref": ref.numpy()}
    cl = ChamferLoss()
    out["emd_approx"] = ref_metrics.emd_approx(sample, ref[:5]).numpy()
    out["earth_mover_distance_b2"] = ref_metrics.earth_mover_distance(sample, ref[:5], batch_size=2).numpy()
    dl, dr = ref_metrics.dist_chamfer(sample, ref[:5], cl)
    out["dist_chamfer_l"], out["dist_chamfer_r"] = dl.numpy(), dr.numpy()
    all_cd, all_emd = ref_metrics._pairwise_EMD_CD_(sample, ref, 3, cl)
    out["pairwise_cd"], out["pairwise_emd"] = all_cd.numpy(), all_emd.numpy()
    for name, v in ref_metrics.mmd_cov(all_cd).items():
        out["mmd_cov_cd__" + name] = np.float32(v)
    res = ref_metrics.compute_all_metrics(sample, ref, 4, cl)
    for name, v in res.items():
        out["all__" + name] = np.float32(v)
    Mxx, Myy = torch.rand(6, 6, generator=g), torch.rand(8, 8, generator=g)
    Mxx, Myy = Mxx + Mxx.t(), Myy + Myy.t()
    Mxy = torch.rand(6, 8, generator=g)
    out["knn_Mxx"], out["knn_Mxy"], out["knn_Myy"] = Mxx.numpy(), Mxy.numpy(), Myy.numpy()
    for name, v in ref_metrics.knn(Mxx, Mxy, Myy, 1).items():
        out["knn1__" + name] = np.float32(v)
    mmd, matched = ref_mmd.minimum_mathing_distance(sample.numpy(), ref.numpy(), 3, device=torch.device("cpu"))
    out["mmd_value"], out["mmd_matched"] = np.float64(mmd), np.array(matched, np.float64)
    np.savez(os.path.join(OUT, "metrics.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
