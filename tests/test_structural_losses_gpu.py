"""GPU parity: HIP structural losses (through the C ABI) vs the CPU oracle and the golden fixtures.

Bars: indices bit-exact; NN distances bit-exact (same fma chain as the oracle); Chamfer scalar
1e-5 relative (north_star); approximate EMD 1e-5 relative on the cost (hardware exp2 vs libm expf —
EMD parity is otherwise unpinned, see oracle/structural_losses_ref.c), per-entry 3e-5 + 1e-3 relative on
`match`, 5e-5 + 1e-3 relative on the cost gradients; every rows-per-lane instance of the EMD sweeps is forced
through hp_emd_set_rows_per_lane and compared with the oracle and, bit for bit, with the others.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    from hyperpocket_amd.utils.pytorch_structural_losses import StructuralLossesBackend
    return StructuralLossesBackend


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _clouds(seed, b, n, m, scale=1.0):
    r = np.random.RandomState(seed)
    return ((r.rand(b, n, 3).astype(np.float32) - 0.5) * scale), ((r.rand(b, m, 3).astype(np.float32) - 0.5) * scale)


# ----------------------------------------------------------------------------- NNDistance
@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 37, 130), (3, 512, 512), (2, 513, 1025), (5, 2048, 2048),
                                    (1, 3000, 700), (70, 64, 64)])
def test_nndistance_bit_exact_vs_oracle(backend, oracle_lib, b, n, m):
    a, c = _clouds(b * 1000 + n, b, n, m)
    d1, i1, d2, i2 = [t.cpu().numpy() for t in backend.NNDistance(_dev(a), _dev(c))]
    r1, j1, r2, j2 = oracle_lib.nndistance(a, c)
    assert np.array_equal(i1, j1) and np.array_equal(i2, j2)
    assert np.array_equal(d1, r1) and np.array_equal(d2, r2)


def test_nndistance_ties_smallest_index(backend, oracle_lib):
    a, c = _clouds(7, 2, 300, 1500)
    c[:, 700:1400] = c[:, 0:700]          # duplicated candidates across LDS tiles
    a[:, 100:200] = a[:, 0:100]
    d1, i1, d2, i2 = [t.cpu().numpy() for t in backend.NNDistance(_dev(a), _dev(c))]
    r1, j1, r2, j2 = oracle_lib.nndistance(a, c)
    assert np.array_equal(i1, j1) and np.array_equal(i2, j2)
    assert (i1 < 700).all() or (j1 >= 1400).any()


@pytest.mark.parametrize("name", ["chamfer_small", "chamfer_ragged", "chamfer_2048"])
def test_nndistance_vs_reference_golden(backend, name):
    g = golden(name)
    d1, i1, d2, i2 = [t.cpu().numpy() for t in backend.NNDistance(_dev(g["preds"]), _dev(g["gts"]))]
    np.testing.assert_allclose(d1, g["dist_pred"], atol=2e-6)
    np.testing.assert_allclose(d2, g["dist_gt"], atol=2e-6)
    assert (i1 == g["idx_pred"]).mean() > 0.999 and (i2 == g["idx_gt"]).mean() > 0.999


def test_nndistance_b_from_first_argument(backend):
    # utils/evaluation/mmd.py:38 passes ref (1,N,3) and chunk (<=64,N,3); b comes from set_d
    a, c = _clouds(3, 1, 128, 128)
    c = np.concatenate([c, c + 0.3, c - 0.2], 0)
    d1, i1, d2, i2 = backend.NNDistance(_dev(a), _dev(c))
    assert d1.shape == (1, 128) and d2.shape == (1, 128)


def test_nndistancegrad_vs_oracle(backend, oracle_lib):
    a, c = _clouds(11, 3, 257, 400)
    A, C = _dev(a), _dev(c)
    d1, i1, d2, i2 = backend.NNDistance(A, C)
    r = np.random.RandomState(5)
    gd1, gd2 = r.randn(3, 257).astype(np.float32), r.randn(3, 400).astype(np.float32)
    g1, g2 = backend.NNDistanceGrad(A, C, i1, i2, _dev(gd1), _dev(gd2))
    o1, o2 = oracle_lib.nndistancegrad(a, c, gd1, i1.cpu().numpy(), gd2, i2.cpu().numpy())
    np.testing.assert_allclose(g1.cpu().numpy(), o1, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(g2.cpu().numpy(), o2, atol=1e-5, rtol=1e-5)


def test_nndistancegrad_clustered_multi_tile_and_deterministic(backend, oracle_lib):
    """Many sources choosing the same few targets (the contended case early in training), more targets than one
    2048-point gradient tile, and bit-identical results run to run (fixed-point LDS accumulation)."""
    r = np.random.RandomState(17)
    a = (r.rand(2, 3000, 3).astype(np.float32) - 0.5)
    c = (r.rand(2, 5000, 3).astype(np.float32) - 0.5) * 3.0
    c[:, :4990] += 5.0                      # only 10 points of c are anywhere near a
    A, C = _dev(a), _dev(c)
    d1, i1, d2, i2 = backend.NNDistance(A, C)
    assert len(np.unique(i1.cpu().numpy())) <= 20
    gd1, gd2 = r.randn(2, 3000).astype(np.float32), r.randn(2, 5000).astype(np.float32)
    g1, g2 = backend.NNDistanceGrad(A, C, i1, i2, _dev(gd1), _dev(gd2))
    j1, j2 = i1.cpu().numpy(), i2.cpu().numpy()
    o1, o2 = oracle_lib.nndistancegrad(a, c, gd1, j1, gd2, j2)
    # the oracle accumulates thousands of O(10) terms sequentially in fp32 (as the reference's atomicAdd does, in some
    # order): loose against it, tight against the same sums in float64
    np.testing.assert_allclose(g1.cpu().numpy(), o1, atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(g2.cpu().numpy(), o2, atol=1e-3, rtol=1e-4)
    t1, t2 = np.zeros(a.shape), np.zeros(c.shape)
    for b in range(2):
        da = 2.0 * gd1[b, :, None].astype(np.float64) * (a[b].astype(np.float64) - c[b][j1[b]])
        dc = 2.0 * gd2[b, :, None].astype(np.float64) * (c[b].astype(np.float64) - a[b][j2[b]])
        t1[b] += da
        np.add.at(t2[b], j1[b], -da)
        t2[b] += dc
        np.add.at(t1[b], j2[b], -dc)
    np.testing.assert_allclose(g1.cpu().numpy(), t1, atol=2e-5, rtol=2e-6)
    np.testing.assert_allclose(g2.cpu().numpy(), t2, atol=2e-5, rtol=2e-6)
    for _ in range(3):
        h1, h2 = backend.NNDistanceGrad(A, C, i1, i2, _dev(gd1), _dev(gd2))
        assert torch.equal(h1, g1) and torch.equal(h2, g2)


def test_nndistancegrad_empty_side(backend):
    A = torch.rand(2, 5, 3, device="cuda")
    C = torch.empty(2, 0, 3, device="cuda")
    i1 = torch.zeros(2, 5, dtype=torch.int32, device="cuda")
    i2 = torch.zeros(2, 0, dtype=torch.int32, device="cuda")
    g1, g2 = backend.NNDistanceGrad(A, C, i1, i2, torch.ones(2, 5, device="cuda"), torch.ones(2, 0, device="cuda"))
    assert g1.shape == (2, 5, 3) and g2.shape == (2, 0, 3) and float(g1.abs().sum()) == 0.0


def test_nn_distance_autograd_function(oracle_lib):
    from hyperpocket_amd.utils.pytorch_structural_losses.nn_distance import nn_distance
    a, c = _clouds(13, 2, 100, 90)
    A, C = _dev(a).requires_grad_(True), _dev(c).requires_grad_(True)
    d1, d2 = nn_distance(A, C)
    (d1.mean(dim=1) + d2.mean(dim=1)).sum().backward()
    r1, j1, r2, j2 = oracle_lib.nndistance(a, c)
    o1, o2 = oracle_lib.nndistancegrad(a, c, np.full_like(r1, 1 / 100), j1, np.full_like(r2, 1 / 90), j2)
    np.testing.assert_allclose(A.grad.cpu().numpy(), o1, atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(C.grad.cpu().numpy(), o2, atol=1e-6, rtol=1e-5)


# ----------------------------------------------------------------------------- ChamferLoss
@pytest.mark.parametrize("name", ["chamfer_small", "chamfer_ragged", "chamfer_2048"])
def test_chamfer_loss_vs_reference_golden(name):
    from hyperpocket_amd.losses.champfer_loss import ChamferLoss
    g = golden(name)
    preds, gts = _dev(g["preds"]).requires_grad_(True), _dev(g["gts"]).requires_grad_(True)
    value = ChamferLoss()(preds, gts)
    assert value.dim() == 0
    assert abs(value.item() - float(g["value"])) <= 1e-5 * abs(float(g["value"]))
    value.backward()
    np.testing.assert_allclose(preds.grad.cpu().numpy(), g["grad_preds"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(gts.grad.cpu().numpy(), g["grad_gts"], atol=1e-5, rtol=1e-4)


def test_chamfer_loss_training_call_convention():
    # core/epoch_loops.py:25-26: loss(gt, reconstruction.permute(0,2,1)) with rec (B,3,N) requiring grad
    from hyperpocket_amd.losses.champfer_loss import ChamferLoss
    a, c = _clouds(17, 4, 256, 256)
    rec = _dev(np.ascontiguousarray(c.transpose(0, 2, 1))).requires_grad_(True)
    gt = _dev(a)
    loss = torch.mean(0.05 * ChamferLoss().cuda()(gt, rec.permute(0, 2, 1)))
    loss.backward()
    assert rec.grad.shape == rec.shape
    ref = torch.from_numpy(c).requires_grad_(True)
    P = ((torch.from_numpy(a)[:, :, None, :] - ref[:, None, :, :]) ** 2).sum(-1)
    l2 = 0.05 * (P.min(1)[0].sum() + P.min(2)[0].sum())
    l2.backward()
    assert abs(loss.item() - l2.item()) <= 1e-5 * abs(l2.item())
    np.testing.assert_allclose(rec.grad.cpu().numpy(), ref.grad.numpy().transpose(0, 2, 1), atol=1e-6, rtol=1e-4)


def test_chamfer_full_size_properties():
    # BASELINE config sizes: B=64, N=2048.  Size-independent properties: symmetry, zero on identical
    # sets, translation invariance, and agreement of the fused sum with the per-point distances.
    from hyperpocket_amd.losses.champfer_loss import ChamferLoss
    from hyperpocket_amd.utils.pytorch_structural_losses import StructuralLossesBackend as B
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(64, 2048, 3, device="cuda", generator=g) - 0.5
    y = torch.rand(64, 2048, 3, device="cuda", generator=g) - 0.5
    L = ChamferLoss()
    lxy, lyx = L(x, y).item(), L(y, x).item()
    assert abs(lxy - lyx) <= 1e-6 * lxy
    assert L(x, x.clone()).item() == 0.0
    assert abs(L(x + 0.25, y + 0.25).item() - lxy) <= 1e-4 * lxy
    d1, i1, d2, i2 = B.NNDistance(x, y)
    assert abs((d1.double().sum() + d2.double().sum()).item() - lxy) <= 1e-6 * lxy
    assert (i1 >= 0).all() and (i1 < 2048).all() and (i2 >= 0).all() and (i2 < 2048).all()
    gathered = torch.gather(y, 1, i1.long().unsqueeze(-1).expand(-1, -1, 3))
    np.testing.assert_allclose(((x - gathered) ** 2).sum(-1).cpu().numpy(), d1.cpu().numpy(), atol=1e-6)


# ----------------------------------------------------------------------------- approximate EMD
@pytest.mark.parametrize("b,n,m", [(2, 64, 64), (3, 200, 200), (2, 300, 150), (1, 130, 390), (2, 1024, 1024), (33, 96, 96)])
def test_approxmatch_vs_oracle(backend, oracle_lib, b, n, m):
    a, c = _clouds(b + n + m, b, n, m)
    match, temp = backend.ApproxMatch(_dev(a), _dev(c))
    assert match.shape == (b, m, n) and temp.shape == (b, 2 * (n + m))
    om, _ = oracle_lib.approxmatch(a, c)                 # KERNEL_CONTRACT: the variant the kernels restate op for op
    got = match.cpu().numpy()
    # hardware exp2 vs libm expf, a summation order of 4 ranges x even/odd halves: per-entry agreement within the envelope two
    # fp32 evaluations of this algorithm have (see _assert_match_close); the cost is the hard gate
    _assert_match_close(got, om)
    cost = backend.MatchCost(_dev(a), _dev(c), match).cpu().numpy()
    ocost = oracle_lib.matchcost(a, c, om)
    np.testing.assert_allclose(cost, ocost, rtol=1e-5)


def test_matchcost_and_grad_vs_oracle_same_match(backend, oracle_lib):
    a, c = _clouds(23, 3, 333, 222)
    om, _ = oracle_lib.approxmatch(a, c)
    A, C, M = _dev(a), _dev(c), _dev(om)
    np.testing.assert_allclose(backend.MatchCost(A, C, M).cpu().numpy(), oracle_lib.matchcost(a, c, om), rtol=2e-6)
    g1, g2 = backend.MatchCostGrad(A, C, M)
    o1, o2 = oracle_lib.matchcostgrad(a, c, om)
    np.testing.assert_allclose(g1.cpu().numpy(), o1, atol=2e-6, rtol=1e-4)
    np.testing.assert_allclose(g2.cpu().numpy(), o2, atol=2e-6, rtol=1e-4)


def test_match_cost_autograd_function(oracle_lib):
    from hyperpocket_amd.utils.pytorch_structural_losses.match_cost import match_cost
    a, c = _clouds(29, 2, 128, 128)
    A, C = _dev(a).requires_grad_(True), _dev(c).requires_grad_(True)
    cost = match_cost(A, C)
    assert cost.shape == (2,)
    w = torch.tensor([0.5, -2.0], device="cuda")
    (cost * w).sum().backward()
    om, _ = oracle_lib.approxmatch(a, c)
    o1, o2 = oracle_lib.matchcostgrad(a, c, om)
    np.testing.assert_allclose(cost.detach().cpu().numpy(), oracle_lib.matchcost(a, c, om), rtol=1e-5)
    wn = w.cpu().numpy()[:, None, None]
    np.testing.assert_allclose(A.grad.cpu().numpy(), o1 * wn, atol=5e-5, rtol=1e-3)
    np.testing.assert_allclose(C.grad.cpu().numpy(), o2 * wn, atol=5e-5, rtol=1e-3)


def test_fused_match_cost_equals_materialised_path(backend):
    """match_cost (match-free: hp_emd_forward/backward) against ApproxMatch -> MatchCost -> MatchCostGrad on the GPU.
    With the level sweeps in the caller's order (hp_emd_set_cull(0)) the two evaluate the same match entries in the same order:
    cost and gradients agree to fp32 rounding of the final sums.  By default (round 6) the match-free path sweeps in Hilbert order
    and leaves the exactly-zero units out — another summation order of the same sums, which the auction amplifies like any other
    fp32 re-ordering: the cost stays inside 2e-6, the gradients inside the envelope every two fp32 evaluations of the algorithm
    share (_assert_grad_close)."""
    from hyperpocket_amd._lib import load_library
    from hyperpocket_amd.utils.pytorch_structural_losses.match_cost import match_cost
    lib = load_library()
    default = lib.hp_emd_set_cull(0)
    try:
        for cull in (0, default):
            lib.hp_emd_set_cull(cull)
            for b, n, m in [(3, 257, 257), (2, 500, 250), (2, 100, 300)]:
                a, c = _clouds(31 + n, b, n, m)
                A, C = _dev(a).requires_grad_(True), _dev(c).requires_grad_(True)
                cost = match_cost(A, C)
                cost.sum().backward()
                match, _ = backend.ApproxMatch(_dev(a), _dev(c))
                cost2 = backend.MatchCost(_dev(a), _dev(c), match)
                g1, g2 = backend.MatchCostGrad(_dev(a), _dev(c), match)
                np.testing.assert_allclose(cost.detach().cpu().numpy(), cost2.cpu().numpy(), rtol=2e-6)
                if cull == 0:
                    np.testing.assert_allclose(A.grad.cpu().numpy(), g1.cpu().numpy(), rtol=1e-5, atol=1e-6)
                    np.testing.assert_allclose(C.grad.cpu().numpy(), g2.cpu().numpy(), rtol=1e-4, atol=2e-6)
                else:
                    _assert_grad_close(A.grad.cpu().numpy(), g1.cpu().numpy(), "grad1, Hilbert-ordered sweeps vs the caller's order")
                    _assert_grad_close(C.grad.cpu().numpy(), g2.cpu().numpy(), "grad2, Hilbert-ordered sweeps vs the caller's order")
    finally:
        lib.hp_emd_set_cull(default)
    # only the second argument needs a gradient in training (match_cost(gt, rec))
    A, C = _dev(a), _dev(c).requires_grad_(True)
    match_cost(A, C).sum().backward()
    assert C.grad is not None


def test_emd_full_size_properties(backend):
    # B=32, N=2048 (BASELINE config 2): mass conservation, identical clouds, permutation equivariance
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(32, 2048, 3, device="cuda", generator=g) - 0.5
    y = torch.rand(32, 2048, 3, device="cuda", generator=g) - 0.5
    match, _ = backend.ApproxMatch(x, y)
    assert (match >= 0).all()
    assert (match.sum(1) <= 1 + 1e-3).all() and (match.sum(2) <= 1 + 1e-3).all()
    assert match.sum().item() / (32 * 2048) > 0.95
    cost = backend.MatchCost(x, y, match)
    perm = torch.randperm(2048, device="cuda", generator=g)
    match_p, _ = backend.ApproxMatch(x[:, perm].contiguous(), y)
    cost_p = backend.MatchCost(x[:, perm].contiguous(), y, match_p)
    np.testing.assert_allclose(cost_p.cpu().numpy(), cost.cpu().numpy(), rtol=1e-4)
    m_same, _ = backend.ApproxMatch(x[:2].contiguous(), x[:2].clone())
    c_same = backend.MatchCost(x[:2].contiguous(), x[:2].clone(), m_same)
    assert (c_same / 2048 < 1e-3).all()


# ----------------------------------------------------------------------------- every EMD template instance
# The packed-record sweeps come in rows-per-lane variants (emd.hip: emd_rows1_kernel<.,.,R>, emd_rows2_kernel<R>,
# emd_grad2_kernel<.,R>) picked by a size heuristic; the headline bench shape (B=64, N=2048) runs R = 2 / 4 / 2.
# hp_emd_set_rows_per_lane forces an instance at any size, so each one meets the oracle at sizes it affords.
@pytest.fixture
def rows_per_lane():
    from hyperpocket_amd._lib import call

    def force(r1, r2, g2):
        call("hp_emd_set_rows_per_lane", r1, r2, g2)
    yield force
    call("hp_emd_set_rows_per_lane", 0, 0, 0)


def _emd_forward(a, c, want1, want2):
    """hp_emd_forward through the C ABI: cost (b,), grad1 / grad2 or None."""
    import ctypes
    from hyperpocket_amd._lib import call, current_stream, load_library
    lib = load_library()
    lib.hp_emd_partials_floats.restype = ctypes.c_long
    A, C = _dev(a), _dev(c)
    b, n, m = A.size(0), A.size(1), C.size(1)
    f32 = dict(dtype=torch.float32, device="cuda")
    temp = torch.empty((b, 2 * (n + m)), **f32)
    ws = torch.empty((lib.hp_approxmatch_workspace_floats(b, n, m),), **f32)
    part = torch.empty((lib.hp_emd_partials_floats(b, n, m),), **f32)
    cost = torch.empty((b,), **f32)
    g1 = torch.empty((b, n, 3), **f32) if want1 else None
    g2 = torch.empty((b, m, 3), **f32) if want2 else None
    call("hp_emd_forward", b, n, m, A, C, temp, ws, part, cost, g1, g2, current_stream(A.device))
    torch.cuda.synchronize()
    return cost, g1, g2


def _assert_match_close(got, want, what="match vs oracle"):
    """Per-entry bar for `match`.  The auction amplifies fp32 rounding: ANY two fp32 evaluations of the algorithm differ in
    a few entries.  Calibration on the C oracle itself (tests/test_oracle_golden.py holds the CPU assertions; numbers from
    (5,200,330), (33,96,96), (512,256,256), (520,200,330)): the contraction variants against each other — max entry
    difference up to 1.8e-3, fraction of entries beyond 3e-5 + 1e-3*|x| up to 5e-5; any variant against its own fp64
    evaluation — max up to 3.5e-3 (33 M entries), fraction up to 1.9e-4.  The kernels are held to that envelope: the
    round-1 bar (3e-5 + 1e-3 relative) for at least 99.97 % of the entries, 5e-3 for every entry; the cost — what
    north_star gates — keeps its 1e-5 (it agrees with exact arithmetic to < 1e-6 for every evaluation).
    The same envelope is applied against the kernels' own contraction variant AND against the literal source
    (contract=0): no per-entry check exists only against the variant that was edited alongside the kernels."""
    err = np.abs(got - want)
    bad = (err > 3e-5 + 1e-3 * np.abs(want)).mean()
    msg = f"{what}: max |err| {err.max():.3e}, fraction beyond 3e-5+1e-3|x| {bad:.3e} of {err.size} entries"
    assert (err <= 5e-3 + 1e-3 * np.abs(want)).all(), msg
    assert bad <= 3e-4, msg
    return msg


def _assert_grad_close(got, want, what="grad vs oracle"):
    """Cost gradients are sums of match entries times unit vectors and inherit the moved entries (oracle variants against
    each other: max component difference up to 2.1e-3, fraction beyond 5e-5 + 1e-3*|x| up to 7.4e-4): the round-1 bar for
    at least 99.8 % of the components, 5e-3 for every one."""
    err = np.abs(got - want)
    bad = (err > 5e-5 + 1e-3 * np.abs(want)).mean()
    msg = f"{what}: max |err| {err.max():.3e}, fraction beyond 5e-5+1e-3|x| {bad:.3e} of {err.size} components"
    assert err.max() < 5e-3, msg
    assert bad <= 2e-3, msg
    return msg


def _oracle_both(oracle_lib, a, c):
    """(match, cost, grad1, grad2) of the C oracle under the contraction the kernels implement (conftest.KERNEL_CONTRACT)
    and under the literal source (contract=0: every product rounded before its add, approxmatch.cu:86-87,131-140,185-189
    as written)."""
    res = []
    for contract in (oracle_lib.KERNEL_CONTRACT, 0):
        om, _ = oracle_lib.approxmatch(a, c, contract=contract)
        res.append((om, oracle_lib.matchcost(a, c, om)) + tuple(oracle_lib.matchcostgrad(a, c, om)))
    return res


EMD_INSTANCES = [(1, 1, 1), (2, 2, 2), (4, 4, 2), (2, 4, 2), (4, 2, 1)]   # (2,4,2) = what B=64, N=2048 selects


@pytest.mark.parametrize("r1,r2,g2", EMD_INSTANCES)
@pytest.mark.parametrize("b,n,m", [(6, 256, 256), (3, 384, 384), (5, 200, 330), (2, 333, 130), (3, 1000, 1000)])
def test_emd_every_rows_per_lane_instance_vs_oracle(backend, oracle_lib, rows_per_lane, b, n, m, r1, r2, g2):
    rows_per_lane(r1, r2, g2)
    a, c = _clouds(b * 7 + n + m, b, n, m)
    (om, ocost, o1, o2), (om_lit, ocost_literal, o1_lit, o2_lit) = _oracle_both(oracle_lib, a, c)
    match, temp = backend.ApproxMatch(_dev(a), _dev(c))
    got = match.cpu().numpy()
    _assert_match_close(got, om, "match vs oracle(contract=3)")
    _assert_match_close(got, om_lit, "match vs oracle(contract=0, literal source)")
    np.testing.assert_allclose(backend.MatchCost(_dev(a), _dev(c), match).cpu().numpy(), ocost, rtol=1e-5)
    # the match-free calls: both gradients; grad2 alone (the cost then rides on the grad2 sweep: core/engine.py)
    cost, g1, g2_ = _emd_forward(a, c, True, True)
    np.testing.assert_allclose(cost.cpu().numpy(), ocost, rtol=1e-5)
    np.testing.assert_allclose(cost.cpu().numpy(), ocost_literal, rtol=1e-5)
    _assert_grad_close(g1.cpu().numpy(), o1, "grad1 vs oracle(contract=3)")
    _assert_grad_close(g2_.cpu().numpy(), o2, "grad2 vs oracle(contract=3)")
    _assert_grad_close(g1.cpu().numpy(), o1_lit, "grad1 vs oracle(contract=0, literal source)")
    _assert_grad_close(g2_.cpu().numpy(), o2_lit, "grad2 (training gradient) vs oracle(contract=0, literal source)")
    cost_b, _, g2_b = _emd_forward(a, c, False, True)
    np.testing.assert_allclose(cost_b.cpu().numpy(), ocost, rtol=1e-5)
    assert torch.equal(g2_b, g2_)
    cost_a, g1_a, _ = _emd_forward(a, c, True, False)
    np.testing.assert_allclose(cost_a.cpu().numpy(), ocost, rtol=1e-5)
    assert torch.equal(g1_a, g1)


@pytest.mark.parametrize("b,n,m", [(512, 256, 256), (520, 200, 330)])
def test_emd_heuristic_picks_the_wide_instances_vs_oracle(backend, oracle_lib, b, n, m):
    """No forcing: at these batch sizes emd.hip's own heuristic selects R = 2 (rows1), 4 (rows2), 2 (grad2) — the
    instances of the bench shape — through the same `pick()` the bench goes through."""
    a, c = _clouds(b + n + m, b, n, m)
    (om, ocost, _, o2), (om_lit, ocost_lit, _, o2_lit) = _oracle_both(oracle_lib, a, c)
    match, _ = backend.ApproxMatch(_dev(a), _dev(c))
    got = match.cpu().numpy()
    _assert_match_close(got, om, "match vs oracle(contract=3)")
    _assert_match_close(got, om_lit, "match vs oracle(contract=0, literal source)")
    cost, _, g2 = _emd_forward(a, c, False, True)
    np.testing.assert_allclose(cost.cpu().numpy(), ocost, rtol=1e-5)
    np.testing.assert_allclose(cost.cpu().numpy(), ocost_lit, rtol=1e-5)
    _assert_grad_close(g2.cpu().numpy(), o2, "grad2 vs oracle(contract=3)")
    _assert_grad_close(g2.cpu().numpy(), o2_lit, "grad2 (training gradient) vs oracle(contract=0, literal source)")


@pytest.mark.parametrize("b,n,m", [(4, 512, 512), (3, 200, 330), (70, 130, 64), (2, 2048, 2048)])
def test_emd_rows_per_lane_instances_agree_bit_for_bit(backend, rows_per_lane, b, n, m):
    """Per row every instance performs the same operations in the same order: match, temp (remainL/R, ratioL/R) and
    the gradients must be IDENTICAL whatever the rows-per-lane setting; the cost is a sum over rows whose grouping into
    workgroup partials follows the setting (fp32 partials, fp64 finish): 2e-6."""
    a, c = _clouds(b + 3 * n + m, b, n, m)
    ref = None
    for r1, r2, g2 in EMD_INSTANCES:
        rows_per_lane(r1, r2, g2)
        match, temp = backend.ApproxMatch(_dev(a), _dev(c))
        cost, g1, g2_ = _emd_forward(a, c, True, True)
        cost_b, _, g2_b = _emd_forward(a, c, False, True)
        got = (match, temp, cost, g1, g2_, cost_b, g2_b)
        if ref is None:
            ref = got
        else:
            for i, (x, y) in enumerate(zip(ref, got)):
                if i in (2, 5):
                    np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-6)
                else:
                    assert torch.equal(x, y), (r1, r2, g2, i)


@pytest.mark.parametrize("b,n,m", [(64, 2048, 2048), (32, 2048, 2048), (33, 1500, 2048), (130, 512, 640)])
def test_emd_two_chains_equal_one_chain(b, n, m):
    """hp_emd_forward runs a batch that fills the chip twice over as two chains of launches on two streams (emd.hip,
    emd_forward_impl).  Clouds are independent and every row sees the same operations in the same order whatever instance its
    half runs: gradients IDENTICAL to the one-chain call's, cost within 2e-6 (the grouping of its partial sums) — with both
    gradients, with the training call's (grad1 = NULL) and at an odd batch (the halves differ in size)."""
    from hyperpocket_amd._lib import load_library
    lib = load_library()
    a, c = _clouds(b + n, b, n, m)
    prev = lib.hp_emd_set_chains(1)
    try:
        one = _emd_forward(a, c, True, True)
        one_b = _emd_forward(a, c, False, True)
        lib.hp_emd_set_chains(2)
        for _ in range(2):      # (twice: the second call re-uses the library's stream and its ordering events)
            two = _emd_forward(a, c, True, True)
            two_b = _emd_forward(a, c, False, True)
            for x, y in ((one, two), (one_b, two_b)):
                np.testing.assert_allclose(y[0].cpu().numpy(), x[0].cpu().numpy(), rtol=2e-6)
                for gx, gy in zip(x[1:], y[1:]):
                    assert (gx is None) == (gy is None)
                    if gx is not None:
                        assert torch.equal(gx, gy)
    finally:
        lib.hp_emd_set_chains(prev)


def _emd_order_tables(ws, b, n, m):
    """What emd_order_kernel leaves in the workspace (emd.hip ws_layout): per cloud the two permutations (position -> caller's
    index), the 8-candidate block boxes and the flag."""
    from hyperpocket_amd._lib import load_library
    per = load_library().hp_approxmatch_workspace_floats(1, n, m)
    w = ws.view(b, per)
    NP, MP = (n + 63) // 64 * 64, (m + 63) // 64 * 64
    off = (NP + 8) * 4 + (MP + 8) * 4 + (MP + 8) + (NP + 8) * 16 + (MP + 8) * 16
    permL = w[:, off:off + NP].view(torch.int32)[:, :n].cpu().numpy()
    permR = w[:, off + NP:off + NP + MP].view(torch.int32)[:, :m].cpu().numpy()
    blkL = w[:, off + NP + MP:off + NP + MP + 6 * (NP // 8)].reshape(b, 6, NP // 8).cpu().numpy()
    flag = w[:, per - 16].cpu().numpy()
    return permL, permR, blkL, flag


def _emd_forward_ws(a, c):
    """hp_emd_forward (grad2 only) keeping the workspace."""
    from hyperpocket_amd._lib import call, current_stream, load_library
    lib = load_library()
    lib.hp_emd_partials_floats.restype = ctypes.c_long
    A, C = _dev(a), _dev(c)
    b, n, m = A.shape[0], A.shape[1], C.shape[1]
    f32 = dict(device=A.device, dtype=torch.float32)
    temp = torch.empty((b, 2 * (n + m)), **f32)
    ws = torch.zeros((lib.hp_approxmatch_workspace_floats(b, n, m),), **f32)
    part = torch.empty((lib.hp_emd_partials_floats(b, n, m),), **f32)
    cost = torch.empty((b,), **f32)
    g2 = torch.full((b, m, 3), float("nan"), **f32)
    call("hp_emd_forward", b, n, m, A, C, temp, ws, part, cost, None, g2, current_stream(A.device))
    torch.cuda.synchronize()
    return cost, g2, ws


@pytest.mark.parametrize("b,n,m", [(3, 96, 96), (2, 200, 330), (5, 330, 200), (2, 1000, 2048), (3, 2048, 2048), (2, 3000, 4096)])
def test_emd_order_kernel_leaves_a_permutation_and_tight_boxes(b, n, m):
    """emd_order_kernel (round 6): the Hilbert order of either set is a permutation of the caller's indices (ragged sizes, sizes
    that are not powers of two, the 4096-point limit), the block boxes are the exact bounding boxes of the 8 points they cover,
    the order is a pure function of the input (two runs: identical tables), and consecutive blocks are compact: the mean block
    diagonal is far below a random block's."""
    from hyperpocket_amd._lib import load_library
    lib = load_library()
    a, c = _clouds(900 + n, b, n, m)
    prev = lib.hp_emd_set_cull(3)
    try:
        _, _, ws = _emd_forward_ws(a, c)
        permL, permR, blkL, flag = _emd_order_tables(ws, b, n, m)
        _, _, ws2 = _emd_forward_ws(a, c)
        assert torch.equal(ws, ws2)                                      # deterministic, records and tables alike
        lib.hp_emd_set_cull(0)
        _, _, ws0 = _emd_forward_ws(a, c)
        assert (_emd_order_tables(ws0, b, n, m)[3] == 0).all()           # the caller's order: flag off
    finally:
        lib.hp_emd_set_cull(prev)
    assert (flag == 1).all()
    for i in range(b):
        assert np.array_equal(np.sort(permL[i]), np.arange(n)) and np.array_equal(np.sort(permR[i]), np.arange(m))
        pts = a[i][permL[i]]
        full = n // 8
        blocks = pts[:full * 8].reshape(full, 8, 3)
        np.testing.assert_array_equal(blkL[i, :3, :full].T, blocks.min(1))
        np.testing.assert_array_equal(blkL[i, 3:, :full].T, blocks.max(1))
        diag = np.linalg.norm(blocks.max(1) - blocks.min(1), axis=1).mean()
        rand = a[i][:full * 8].reshape(full, 8, 3)
        assert diag < (0.5 if n >= 512 else 0.75) * np.linalg.norm(rand.max(1) - rand.min(1), axis=1).mean()


@pytest.mark.parametrize("b,n,m", [(3, 96, 96), (2, 200, 330), (5, 330, 200), (66, 2048, 2048), (1, 4100, 64)])
def test_emd_culling_sweeps_equal_the_full_sweeps_in_the_same_order(b, n, m):
    """The culling sweeps leave out (64-row tile, 8-candidate block) units whose every exponential is exactly zero.  With the
    records in the same (Hilbert) order, culling NO level (hp_emd_set_cull(k), k levels culled) and culling 1, 3, 4 or all 9
    levels must therefore give the same cost and gradients up to the candidate-range grouping of the partial sums (the culling
    kernels interleave the candidate blocks over a workgroup's four waves, the plain ones give each wave a contiguous quarter):
    cost within 2e-6, gradients inside the fp32 re-ordering envelope.  And against the caller's order (cull 0): the same bars.
    Clouds in the +-0.5 cube: at levels -16384 .. -1024 most units are beyond the underflow radius, so the skipping is exercised.
    (1, 4100, 64): a set past the ordering kernel's 4096-point limit — the call falls back to the caller's order, all settings alike.)"""
    from hyperpocket_amd._lib import load_library
    lib = load_library()
    a, c = _clouds(77 + n, b, n, m)
    prev = lib.hp_emd_set_cull(0)
    try:
        base = _emd_forward(a, c, True, True)
        for k in (1, 3, 4, 9):
            lib.hp_emd_set_cull(k)
            got = _emd_forward(a, c, True, True)
            assert all(torch.isfinite(t).all() for t in got)
            np.testing.assert_allclose(got[0].cpu().numpy(), base[0].cpu().numpy(), rtol=2e-6, err_msg=f"cost, cull={k}")
            _assert_grad_close(got[1].cpu().numpy(), base[1].cpu().numpy(), f"grad1, cull={k} vs caller's order")
            _assert_grad_close(got[2].cpu().numpy(), base[2].cpu().numpy(), f"grad2, cull={k} vs caller's order")
            again = _emd_forward(a, c, True, True)
            assert all(torch.equal(x, y) for x, y in zip(got, again)), f"cull={k}: not reproducible"
    finally:
        lib.hp_emd_set_cull(prev)


def test_emd_culling_far_apart_sets_and_degenerate_clouds(oracle_lib):
    """Edge cases of the ordering / culling path against the oracle: (1) the two sets far apart (every unit beyond every culling
    radius: all masks empty, ratioL = remainL / 1e-9 as in approxmatch.cu:89-93), (2) every point of a set identical (one Hilbert
    cell, degenerate boxes), (3) a set with n = 1."""
    r = np.random.RandomState(3)
    a = (r.rand(2, 300, 3).astype(np.float32) - 0.5)
    far = a[:, :200] + np.float32(3.0)
    same = np.repeat(r.rand(2, 1, 3).astype(np.float32), 256, 1)
    for x, y in ((a, far), (same, a), (a, same), (a[:, :1], a)):
        cost, g1, g2 = _emd_forward(x, y, True, True)
        om, _ = oracle_lib.approxmatch(x, y)
        np.testing.assert_allclose(cost.cpu().numpy(), oracle_lib.matchcost(x, y, om), rtol=1e-5, atol=1e-6)
        o1, o2 = oracle_lib.matchcostgrad(x, y, om)
        _assert_grad_close(g1.cpu().numpy(), o1)
        _assert_grad_close(g2.cpu().numpy(), o2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: a stream of cuda:1 while cuda:0 is the current device")
def test_emd_forward_on_a_stream_of_another_device_than_the_current_one():
    """hp_emd_forward runs half of the clouds on a stream the library owns.  That stream must live on the device of the
    CALLER'S stream, not on whatever device the calling thread has current (emd.hip chain_stream): tensors and stream on cuda:1,
    current device cuda:0, result equal to the same call made with cuda:1 current."""
    from hyperpocket_amd._lib import call, load_library
    lib = load_library()
    lib.hp_emd_partials_floats.restype = ctypes.c_long
    a, c = _clouds(12, 66, 2048, 2048)
    dev = torch.device("cuda", 1)
    A, C = torch.from_numpy(a).to(dev), torch.from_numpy(c).to(dev)
    b, n, m = A.shape[0], A.shape[1], C.shape[1]
    f32 = dict(device=dev, dtype=torch.float32)

    def run():
        temp = torch.empty((b, 2 * (n + m)), **f32)
        ws = torch.empty((lib.hp_approxmatch_workspace_floats(b, n, m),), **f32)
        part = torch.empty((lib.hp_emd_partials_floats(b, n, m),), **f32)
        cost, g2 = torch.empty((b,), **f32), torch.empty((b, m, 3), **f32)
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        call("hp_emd_forward", b, n, m, A, C, temp, ws, part, cost, None, g2, ctypes.c_void_p(st.cuda_stream))
        st.synchronize()
        return cost.cpu(), g2.cpu()
    with torch.cuda.device(1):
        want = run()
    with torch.cuda.device(0):
        got = run()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_emd_final_sweep_derived_exponentials_vs_all_nine_from_hardware(oracle_lib):
    """The match-free cost / gradient sweep forms four of its nine per-level exponentials as the fourth power of their
    neighbour's (emd.hip match_entry2<., DERIVE>; default on).  Against the same sweep with all nine from v_exp_f32, over the
    regimes of the cost error map at N = 2048 plus ragged small shapes: cost within 1e-6 relative, both gradients within 2e-6
    of their scale; and the exact form still equals the materialising path's cost to its old bar.  (Against the oracle the
    derived form is what test_emd_cost_error_distribution_at_full_size and the full-size call tests measure, at unchanged bars.)"""
    from hyperpocket_amd._lib import load_library
    lib = load_library()
    cases = [(_clouds(5, 3, 200, 330)), (_clouds(6, 5, 64, 64))]
    for gt, rec in _emd_regimes().values():
        cases.append((gt[:6], rec[:6]))
    worst_c = worst_g = 0.0
    prev = lib.hp_emd_set_final_derive(1)
    try:
        for a, c in cases:
            lib.hp_emd_set_final_derive(1)
            cost_d, g1_d, g2_d = _emd_forward(a, c, True, True)
            cost_b, _, g2_b = _emd_forward(a, c, False, True)          # the training call's shape: cost rides on the grad2 sweep
            lib.hp_emd_set_final_derive(0)
            cost_e, g1_e, g2_e = _emd_forward(a, c, True, True)
            assert torch.equal(g2_d, g2_b)
            mass = cost_e > 1e-3
            rel = ((cost_d - cost_e).abs() / cost_e.abs().clamp_min(1e-30))[mass]
            relb = ((cost_b - cost_e).abs() / cost_e.abs().clamp_min(1e-30))[mass]
            assert rel.numel() == 0 or max(rel.max().item(), relb.max().item()) <= 1e-6, (rel.max().item(), relb.max().item())
            assert ((cost_d - cost_e).abs()[~mass] <= 1e-6).all()
            for gd, ge in ((g1_d, g1_e), (g2_d, g2_e)):
                scale = ge.abs().max().item()
                err = (gd - ge).abs().max().item()
                assert err <= 2e-6 * scale + 1e-12, (err, scale)
                worst_g = max(worst_g, err / max(scale, 1e-30))
            if rel.numel():
                worst_c = max(worst_c, rel.max().item())
        print(f"\nderived vs exact final sweep: worst cost rel diff {worst_c:.2e}, worst gradient diff / scale {worst_g:.2e}")
    finally:
        lib.hp_emd_set_final_derive(prev)


@pytest.mark.parametrize("B", [64, 32])
def test_emd_training_call_full_size_vs_oracle(oracle_lib, B):
    """The call core/engine.py makes at the bench shape — hp_emd_forward(B=64, N=2048, grad1=NULL, grad2 != NULL), i.e.
    emd_rows1_kernel<.,.,2>, emd_rows2_kernel<4>, emd_grad2_kernel<true,2> — and at BASELINE configs[1]'s own batch
    (B=32: pick() selects (2,2,.) there) against the oracle on 4 of the clouds (uniform gt vs uniform rec, and gt vs a
    noisy copy of itself: late-training geometry).  Cost AND the training gradient grad2 against both the kernels'
    contraction variant and the literal source (contract=0)."""
    r = np.random.RandomState(B)
    gt = r.rand(B, 2048, 3).astype(np.float32) - 0.5
    rec = r.rand(B, 2048, 3).astype(np.float32) - 0.5
    h = B // 2
    rec[h:] = gt[h:][:, r.permutation(2048)] + 0.02 * r.randn(B - h, 2048, 3).astype(np.float32)
    cost, _, g2 = _emd_forward(gt, rec, False, True)
    cost, g2 = cost.cpu().numpy(), g2.cpu().numpy()
    assert np.isfinite(cost).all() and np.isfinite(g2).all()
    pick = [0, B // 4 + 1, h + B // 8, B - 1]
    (om, ocost, _, o2), (om0, ocost0, _, o2_lit) = _oracle_both(oracle_lib, gt[pick], rec[pick])
    np.testing.assert_allclose(cost[pick], ocost, rtol=1e-5)
    np.testing.assert_allclose(cost[pick], ocost0, rtol=1e-5)      # the literal (uncontracted) source: same gate
    _assert_grad_close(g2[pick], o2, "grad2 vs oracle(contract=3)")
    _assert_grad_close(g2[pick], o2_lit, "grad2 (training gradient) vs oracle(contract=0, literal source)")


@pytest.mark.parametrize("b,n,m,side", [(3, 200, 200, True), (2, 1024, 1024, False), (64, 512, 512, True)])
def test_emd_forward_acc_adds_onto_a_running_gradient(b, n, m, side):
    """hp_emd_forward_acc (the engine's call): grad2_acc += scale * grad2 inside the gradient sweep, ordered behind the stream
    that wrote the running gradient — equals hp_emd_forward followed by the axpy it replaces (to the rounding of one fma)."""
    from hyperpocket_amd._lib import call, current_stream, load_library
    import ctypes
    a, c = _clouds(7 * b + n, b, n, m)
    cost0, _, g2 = _emd_forward(a, c, False, True)
    A, C = _dev(a), _dev(c)
    lib = load_library()
    lib.hp_emd_partials_floats.restype = ctypes.c_long
    f32 = dict(dtype=torch.float32, device="cuda")
    temp = torch.empty((b, 2 * (n + m)), **f32)
    ws = torch.empty((max(1, lib.hp_approxmatch_workspace_floats(b, n, m)),), **f32)
    part = torch.empty((max(1, lib.hp_emd_partials_floats(b, n, m)),), **f32)
    cost = torch.empty((b,), **f32)
    scale = 0.05 / m
    s2 = torch.cuda.Stream() if side else None
    cur = torch.cuda.current_stream()
    if side:
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            torch.cuda._sleep(2_000_000)          # the running gradient lands late on the other stream
            acc = torch.randn(b, m, 3, **f32)
            base = acc.clone()
    else:
        acc = torch.randn(b, m, 3, **f32)
        base = acc.clone()
    call("hp_emd_forward_acc", b, n, m, A, C, temp, ws, part, cost, acc, float(scale), current_stream(A.device),
         ctypes.c_void_p(s2.cuda_stream if side else 0))
    torch.cuda.synchronize()
    assert torch.equal(cost, cost0)
    want = base.double() + scale * g2.double()
    err = (acc.double() - want).abs().max().item()
    assert err <= 1e-6 * max(1.0, want.abs().max().item()), err


# ----------------------------------------------------------------------------- the reference's exact launcher prototypes
def _exact_approxmatch(a, c):
    """hp_approxmatch(b,n,m,xyz1,xyz2,match,temp,stream) — structural_loss.cpp:11's argument list, nothing else."""
    from hyperpocket_amd._lib import call, current_stream
    A, C = _dev(a), _dev(c)
    b, n, m = A.size(0), A.size(1), C.size(1)
    match = torch.full((b, m, n), float("nan"), device="cuda")     # torch::empty in the reference binding
    temp = torch.full((b, 2 * (n + m)), float("nan"), device="cuda")
    call("hp_approxmatch", b, n, m, A, C, match, temp, current_stream(A.device))
    return match, temp


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 64, 64), (3, 200, 200), (2, 300, 150), (1, 130, 390), (2, 1024, 1024),
                                    (33, 96, 96), (2, 1500, 7)])
def test_exact_prototype_approxmatch_and_matchcost_vs_oracle(backend, oracle_lib, b, n, m):
    from hyperpocket_amd._lib import call, current_stream
    a, c = _clouds(b + n + 2 * m, b, n, m)
    match, temp = _exact_approxmatch(a, c)
    om, otemp = oracle_lib.approxmatch(a, c)
    # same summation order as the oracle (one accumulator, ascending candidates): only v_exp_f32 vs exp2f differs
    _assert_match_close(match.cpu().numpy(), om)
    # temp = [remainL | remainR | ratioL | ratioR]: the remaining masses are comparable; the last level's ratios are
    # remain / (1e-9 + ~0), i.e. rounding residue of the masses times 1e9 — finite, not comparable
    t = temp.cpu().numpy()
    assert np.isfinite(t).all()
    np.testing.assert_allclose(t[:, :n + m], otemp[:, :n + m], atol=1e-5)
    out = torch.full((b,), float("nan"), device="cuda")
    call("hp_matchcost", b, n, m, _dev(a), _dev(c), match, out, current_stream(match.device))   # structural_loss.cpp:12
    np.testing.assert_allclose(out.cpu().numpy(), oracle_lib.matchcost(a, c, om), rtol=1e-5)
    # ... and the scratch-taking fast variants the Python binding uses give the same matching
    match_ws, _ = backend.ApproxMatch(_dev(a), _dev(c))
    _assert_match_close(match_ws.cpu().numpy(), match.cpu().numpy())
    np.testing.assert_allclose(backend.MatchCost(_dev(a), _dev(c), match).cpu().numpy(), out.cpu().numpy(), rtol=2e-6)


def test_exact_prototype_approxmatch_full_size(backend):
    """B=32, N=2048 through the workspace-free launcher: equals the record-based path up to fp32 summation order."""
    a, c = _clouds(5, 32, 2048, 2048)
    match, _ = _exact_approxmatch(a, c)
    match_ws, _ = backend.ApproxMatch(_dev(a), _dev(c))
    assert torch.isfinite(match).all()
    assert (match - match_ws).abs().max().item() < 2e-3          # single entries move with the summation order ...
    c1 = backend.MatchCost(_dev(a), _dev(c), match).cpu().numpy()
    c2 = backend.MatchCost(_dev(a), _dev(c), match_ws).cpu().numpy()
    np.testing.assert_allclose(c1, c2, rtol=1e-5)                # ... the cost does not


# ----------------------------------------------------------------------------- distance from exact arithmetic
@pytest.mark.parametrize("b,n,m", [(4, 256, 256), (4, 512, 512), (4, 300, 150), (2, 1024, 1024)])
def test_emd_distance_from_fp64_evaluation(backend, oracle_lib, b, n, m):
    """The EMD half of the oracle is parity-unpinned (no reference build, no vectors).  What CAN be bounded: the distance
    of the HIP kernels from the fp64 evaluation of the same nine-level algorithm, next to the distance of the fp32 C
    oracle under every fma-contraction assumption (tests/test_oracle_golden.py holds the CPU half).  Single match
    entries move by ~1e-4 between ANY two fp32 evaluations (the auction amplifies rounding); the cost — the quantity
    north_star gates at 1e-5 — agrees with exact arithmetic to ~1e-6."""
    a, c = _clouds(b + n + m, b, n, m)
    m64, c64 = oracle_lib.approxmatch_f64(a, c)
    match, _ = backend.ApproxMatch(_dev(a), _dev(c))
    cost = backend.MatchCost(_dev(a), _dev(c), match).cpu().numpy()
    far = lambda x: (np.abs(x - m64) > 3e-5 + 1e-3 * np.abs(m64)).mean()     # noqa: E731
    assert far(match.cpu().numpy()) <= 3e-4
    np.testing.assert_allclose(cost, c64, rtol=2e-6)
    cost_mf, _, _ = _emd_forward(a, c, False, True)
    np.testing.assert_allclose(cost_mf.cpu().numpy(), c64, rtol=2e-6)
    mx, _ = _exact_approxmatch(a, c)
    assert far(mx.cpu().numpy()) <= 3e-4


def test_nndistancegrad_tiny_upstream_gradients(backend):
    """Mean-reduced callers pass grad_dist ~ 1/(B*N) ~ 1e-6: the scatter half's fixed-point grid scales with the data
    (structural_losses.hip), so small gradients keep fp32-level relative accuracy."""
    r = np.random.RandomState(3)
    a = r.rand(2, 700, 3).astype(np.float32) - 0.5
    c = (r.rand(2, 900, 3).astype(np.float32) - 0.5) * 0.05           # clustered: many sources share a target
    A, C = _dev(a), _dev(c)
    d1, i1, d2, i2 = backend.NNDistance(A, C)
    for mag in (1e-6, 1e-12, 1e4):
        gd1 = (r.rand(2, 700).astype(np.float32) + 0.5) * mag
        gd2 = (r.rand(2, 900).astype(np.float32) + 0.5) * mag
        g1, g2 = backend.NNDistanceGrad(A, C, i1, i2, _dev(gd1), _dev(gd2))
        j1, j2 = i1.cpu().numpy(), i2.cpu().numpy()
        t1, t2 = np.zeros(a.shape), np.zeros(c.shape)
        for b in range(2):
            da = 2.0 * gd1[b, :, None].astype(np.float64) * (a[b].astype(np.float64) - c[b][j1[b]])
            dc = 2.0 * gd2[b, :, None].astype(np.float64) * (c[b].astype(np.float64) - a[b][j2[b]])
            t1[b] += da
            np.add.at(t2[b], j1[b], -da)
            t2[b] += dc
            np.add.at(t1[b], j2[b], -dc)
        np.testing.assert_allclose(g1.cpu().numpy(), t1, rtol=3e-6, atol=3e-7 * mag)
        np.testing.assert_allclose(g2.cpu().numpy(), t2, rtol=3e-6, atol=3e-7 * mag)


def test_inputs_are_validated(backend):
    a, c = _clouds(1, 1, 8, 8)
    from hyperpocket_amd import HipExtensionError
    with pytest.raises(RuntimeError):
        backend.NNDistance(torch.from_numpy(a), torch.from_numpy(c))          # CPU tensors: no CPU path
    with pytest.raises(HipExtensionError):
        backend.NNDistance(_dev(a).transpose(1, 2), _dev(c))                  # non-contiguous


# ----------------------------------------------------------------------------- the EMD cost error, mapped
def _emd_regimes():
    """>= 200 clouds at the metric's N = 2048 in the regimes a training run visits: (gt, rec) pairs."""
    r = np.random.RandomState(20260)
    N = 2048
    out = {}
    gt = r.rand(56, N, 3).astype(np.float32) - 0.5
    out["uniform vs uniform"] = (gt, r.rand(56, N, 3).astype(np.float32) - 0.5)
    gt = r.rand(56, N, 3).astype(np.float32) - 0.5
    sig = np.repeat(np.array([0.002, 0.01, 0.02, 0.05], np.float32), 14)[:, None, None]
    perm = np.stack([g[r.permutation(N)] for g in gt])
    out["noisy copy (late training)"] = (gt, perm + sig * r.randn(56, N, 3).astype(np.float32))
    # clustered: mixtures of 3..12 tight Gaussians (object parts), rec a different draw from the same mixture
    gts, recs = [], []
    for i in range(48):
        k = 3 + i % 10
        cen = (r.rand(k, 3) - 0.5) * 0.8
        s = 0.01 + 0.05 * r.rand(k, 1)
        za, zb = r.randint(0, k, N), r.randint(0, k, N)
        gts.append(cen[za] + s[za] * r.randn(N, 3))
        recs.append(cen[zb] + s[zb] * r.randn(N, 3))
    out["clustered"] = (np.asarray(gts, np.float32), np.asarray(recs, np.float32))
    return out


def _untrained_network_regime(B=48):
    """gt in the +-0.5 cube against the output of an UNTRAINED xavier-sqrt2 HyperPocket (rec at O(10^2)): the operating point
    of tests/test_model_gpu.py::test_baseline_config2_config3_per_gpu_step, where one comparison once sat at 1.16e-5."""
    import copy
    from hyperpocket_amd import ops
    from hyperpocket_amd.core.setup import weights_init
    from hyperpocket_amd.model.full_model import FullModel
    cfg = {"random_encoder": {"output_size": 128, "use_bias": True, "relu_slope": 0.2},
           "real_encoder": {"output_size": 128, "use_bias": True, "relu_slope": 0.2},
           "hyper_network": {"use_bias": True, "relu_slope": 0.2},
           "target_network": {"use_bias": True, "relu_slope": 0.2, "freeze_layers_learning": False,
                              "layer_out_channels": [32, 64, 128, 64]},
           "target_network_input": {"constant": False, "normalization": {"enable": True, "type": "progressive", "epoch": 100}}}
    torch.manual_seed(2020)
    model = FullModel(copy.deepcopy(cfg))
    model.apply(weights_init)
    model = model.cuda().train()
    g = torch.Generator().manual_seed(77)
    ex, mi = torch.rand(B, 1024, 3, generator=g) - 0.5, torch.rand(B, 1024, 3, generator=g) - 0.5
    gt = torch.cat([ex, mi], 1)
    try:
        with torch.no_grad():
            rec, _, _ = model(ex.clone().cuda(), mi.clone().cuda(), [B, 2048, 3], 1, torch.device("cuda"))
        rec = rec.permute(0, 2, 1).contiguous().cpu().numpy()
    finally:
        ops.clear_grad_views()
    return gt.numpy(), rec


def test_emd_cost_error_distribution_at_full_size(oracle_lib):
    """The EMD cost north_star gates at 1e-5, MAPPED instead of sampled: hp_emd_forward (the engine's call, B = regime
    size, N = 2048) against the C oracle under the kernels' contraction (3) and the literal source (0) on 208 clouds in four
    regimes.  Asserted: every cloud whose cost carries mass (> 1e-3; a matched cloud's is O(10..1000)) within 1e-5
    relative (the gate), 99th percentile within 2e-6; the distribution is printed (pytest -s) and quoted in docs/DESIGN_HISTORY.md 2.
    Measured in round 4 (MI355X): max 7.6e-7, p99 <= 6.4e-7, median 0..1.6e-7 in every regime and under both contractions —
    at the untrained-network operating point (where docs/DESIGN_HISTORY.md 3.6 once saw 1.16e-5 with another MFMA shape upstream) the 15 of
    48 clouds that carry mass sit at <= 6.5e-7 with the shipped kernels; the other 33 have cost ~1e-24 (every exponential
    underflows) and are held to an absolute 1e-6."""
    regimes = _emd_regimes()
    regimes["untrained network (rec O(1e2))"] = _untrained_network_regime()
    total, lines = 0, []
    worst = 0.0
    for name, (gt, rec) in regimes.items():
        cost, _, g2 = _emd_forward(gt, rec, False, True)
        cost = cost.cpu().numpy().astype(np.float64)
        assert np.isfinite(cost).all() and torch.isfinite(g2).all(), name
        for contract in (oracle_lib.KERNEL_CONTRACT, 0):
            om, _ = oracle_lib.approxmatch(gt, rec, contract=contract)
            want = oracle_lib.matchcost(gt, rec, om).astype(np.float64)
            mass = want > 1e-3
            rel = np.abs(cost[mass] - want[mass]) / want[mass]
            # clouds without mass: every exponential underflowed (cost ~1e-24): absolute agreement only
            assert np.all(np.abs(cost[~mass] - want[~mass]) <= 1e-6), (name, contract)
            if rel.size:
                p50, p99, mx = np.percentile(rel, 50), np.percentile(rel, 99), rel.max()
                worst = max(worst, mx)
                lines.append(f"{name:32s} contract={contract}: {int(mass.sum()):3d}/{len(want)} clouds with mass, cost "
                             f"{want[mass].min():.3g}..{want[mass].max():.3g}, rel err median {p50:.2e} p99 {p99:.2e} max {mx:.2e}")
                assert mx <= 1e-5, lines[-1]
                assert p99 <= 2e-6, lines[-1]
            else:
                lines.append(f"{name:32s} contract={contract}: no cloud with mass (all {len(want)} costs < 1e-3)")
        total += len(gt)
    assert total >= 200
    print("\nEMD cost error map (hp_emd_forward vs oracle), N=2048:\n" + "\n".join(lines))
