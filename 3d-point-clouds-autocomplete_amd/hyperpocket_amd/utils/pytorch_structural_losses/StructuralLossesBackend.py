"""Drop-in for the reference's compiled module ``StructuralLossesBackend``
(utils/pytorch_structural_losses/structural_loss.cpp:130-136): the same five functions with the
same argument order, output shapes/dtypes and `b, n` taken from ``set_d``, `m` from ``set_q``
(SURVEY Q12 — no silent broadcasting), bound to the HIP C ABI instead of the CUDA launchers.  A binding may allocate, so ApproxMatch / MatchCost call the
scratch-taking fast variants (hp_approxmatch_ws / hp_matchcost_ws); the launchers with the reference's exact argument
lists (hp_approxmatch / hp_matchcost, structural_loss.cpp:11-12) are what INTEGRATION.md §2 rebinds and what
tests/test_structural_losses_gpu.py calls through ctypes.
"""
import torch

from ..._lib import call, check_input, current_stream, load_library


def _dims(set_d, set_q):
    return set_d.size(0), set_d.size(1), set_q.size(1)


def ApproxMatch(set_d, set_q):
    """structural_loss.cpp:26-41 -> [match (b,m,n), temp (b,2(n+m))]"""
    check_input(set_d, "set_d")
    check_input(set_q, "set_q")
    b, n, m = _dims(set_d, set_q)
    match = torch.empty((b, m, n), dtype=torch.float32, device=set_d.device)
    temp = torch.empty((b, (n + m) * 2), dtype=torch.float32, device=set_d.device)
    ws = torch.empty((max(1, load_library().hp_approxmatch_workspace_floats(b, n, m)),), dtype=torch.float32,
                     device=set_d.device)
    call("hp_approxmatch_ws", b, n, m, set_d, set_q, match, temp, ws, current_stream(set_d.device))
    return [match, temp]


def MatchCost(set_d, set_q, match):
    """structural_loss.cpp:43-56 -> out (b,)"""
    check_input(set_d, "set_d")
    check_input(set_q, "set_q")
    check_input(match, "match")
    b, n, m = _dims(set_d, set_q)
    out = torch.empty((b,), dtype=torch.float32, device=set_d.device)
    part = torch.empty((max(1, load_library().hp_matchcost_workspace_floats(b, n, m)),), dtype=torch.float32,
                       device=set_d.device)
    call("hp_matchcost_ws", b, n, m, set_d, set_q, match, out, part, current_stream(set_d.device))
    return out


def MatchCostGrad(set_d, set_q, match):
    """structural_loss.cpp:58-73 -> [grad1 (b,n,3), grad2 (b,m,3)]"""
    check_input(set_d, "set_d")
    check_input(set_q, "set_q")
    check_input(match, "match")
    b, n, m = _dims(set_d, set_q)
    grad1 = torch.empty((b, n, 3), dtype=torch.float32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=torch.float32, device=set_d.device)
    call("hp_matchcostgrad", b, n, m, set_d, set_q, match, grad1, grad2, current_stream(set_d.device))
    return [grad1, grad2]


def NNDistance(set_d, set_q):
    """structural_loss.cpp:84-103 -> [dist1 (b,n), idx1 (b,n) int32, dist2 (b,m), idx2 (b,m) int32]"""
    check_input(set_d, "set_d")
    check_input(set_q, "set_q")
    b, n, m = _dims(set_d, set_q)
    dev = set_d.device
    dist1 = torch.empty((b, n), dtype=torch.float32, device=dev)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=dev)
    dist2 = torch.empty((b, m), dtype=torch.float32, device=dev)
    idx2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    call("hp_nndistance", b, n, set_d, m, set_q, dist1, idx1, dist2, idx2, current_stream(dev))
    return [dist1, idx1, dist2, idx2]


def NNDistanceGrad(set_d, set_q, idx1, idx2, grad_dist1, grad_dist2):
    """structural_loss.cpp:105-128 -> [grad1 (b,n,3), grad2 (b,m,3)]"""
    check_input(set_d, "set_d")
    check_input(set_q, "set_q")
    check_input(idx1, "idx1", torch.int32)
    check_input(idx2, "idx2", torch.int32)
    check_input(grad_dist1, "grad_dist1")
    check_input(grad_dist2, "grad_dist2")
    b, n, m = _dims(set_d, set_q)
    grad1 = torch.empty((b, n, 3), dtype=torch.float32, device=set_d.device)
    grad2 = torch.empty((b, m, 3), dtype=torch.float32, device=set_d.device)
    call("hp_nndistancegrad", b, n, set_d, m, set_q, grad_dist1, idx1, grad_dist2, idx2, grad1, grad2,
         current_stream(set_d.device))
    return [grad1, grad2]
