"""GPU parity of the evaluation consumers against fixtures produced by the reference's own host
code with the CPU oracle standing in for its compiled backend (tests/golden/make_golden_metrics.py)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return golden("metrics")


def _c(a):
    return torch.from_numpy(a).cuda()


def test_emd_and_chamfer_wrappers(g):
    from hyperpocket_amd.utils import metrics as M
    s, r = _c(g["sample"]), _c(g["ref"])
    np.testing.assert_allclose(M.emd_approx(s, r[:5]).cpu().numpy(), g["emd_approx"], rtol=1e-5)
    np.testing.assert_allclose(M.earth_mover_distance(s, r[:5], batch_size=2).cpu().numpy(), g["earth_mover_distance_b2"], rtol=1e-5)
    dl, dr = M.dist_chamfer(s, r[:5])
    np.testing.assert_allclose(dl.cpu().numpy(), g["dist_chamfer_l"], atol=2e-6)
    np.testing.assert_allclose(dr.cpu().numpy(), g["dist_chamfer_r"], atol=2e-6)
    res = M.EMD_CD(s, r[:5], 2, reduced=False)
    np.testing.assert_allclose(res["MMD-EMD"].cpu().numpy(), g["emd_approx"], rtol=1e-5)
    np.testing.assert_allclose(res["MMD-CD"].cpu().numpy(), g["dist_chamfer_l"].mean(1) + g["dist_chamfer_r"].mean(1), rtol=1e-5)


def test_pairwise_and_summary_metrics(g):
    from hyperpocket_amd.utils import metrics as M
    s, r = _c(g["sample"]), _c(g["ref"])
    cd, emd = M._pairwise_EMD_CD_(s, r, 3)
    assert cd.shape == (5, 7) and emd.shape == (5, 7)
    np.testing.assert_allclose(cd.cpu().numpy(), g["pairwise_cd"], rtol=1e-5)
    np.testing.assert_allclose(emd.cpu().numpy(), g["pairwise_emd"], rtol=1e-5)
    for k, v in M.mmd_cov(cd).items():
        assert abs(v.item() - float(g["mmd_cov_cd__" + k])) <= 1e-5 * max(1.0, abs(float(g["mmd_cov_cd__" + k]))), k
    for k, v in M.compute_all_metrics(s, r, 4).items():
        assert abs(v.item() - float(g["all__" + k])) <= 1e-5 * max(1.0, abs(float(g["all__" + k]))), k


def test_knn_two_sample_test(g):
    from hyperpocket_amd.utils import metrics as M
    res = M.knn(_c(g["knn_Mxx"]), _c(g["knn_Mxy"]), _c(g["knn_Myy"]), 1)
    for k, v in res.items():
        assert abs(v.item() - float(g["knn1__" + k])) <= 1e-6, k


def test_minimum_matching_distance_keeps_reference_semantics(g):
    from hyperpocket_amd.utils.evaluation.mmd import minimum_mathing_distance
    mmd, matched = minimum_mathing_distance(g["sample"], g["ref"], 3, device=torch.device("cuda"))
    np.testing.assert_allclose(np.array(matched), g["mmd_matched"], rtol=1e-5)
    assert abs(mmd - float(g["mmd_value"])) <= 1e-5 * float(g["mmd_value"])
    with pytest.raises(ValueError):
        minimum_mathing_distance(g["sample"][:, :50], g["ref"], 3, device=torch.device("cuda"))
