"""Evaluation consumers of the structural losses — the EMD/CD part of the reference's
utils/metrics.py:44-241, same names, arguments and results, over the HIP kernels.

What is different underneath: Chamfer distances come from the fused nearest-neighbour kernel
(hp_nndistance) instead of the (B,N,N) `batch_pairwise_dist` tensor, EMD from the match-free path
(hp_emd_forward).  The JSD / occupancy-grid helpers of the reference file are CPU numpy code with no kernel
behind them and are out of scope.
"""
import torch

from .pytorch_structural_losses.match_cost import match_cost
from .pytorch_structural_losses.nn_distance import nn_distance


def emd_approx(sample, ref):
    """utils/metrics.py:71-76 — per-cloud approximate EMD divided by the number of points."""
    n, n_ref = sample.size(1), ref.size(1)
    assert n == n_ref, "Not sure what would EMD do in this case"
    return match_cost(sample.contiguous(), ref.contiguous()) / float(n)


def earth_mover_distance(sample_pcs, ref_pcs, batch_size=None):
    """utils/metrics.py:44-68"""
    sample_pcs, ref_pcs = sample_pcs.contiguous(), ref_pcs.contiguous()
    if sample_pcs.dim() == 2:
        sample_pcs = sample_pcs.unsqueeze(0)
    if ref_pcs.dim() == 2:
        ref_pcs = ref_pcs.unsqueeze(0)
    n_sample, n_ref = sample_pcs.shape[0], ref_pcs.shape[0]
    assert n_sample == n_ref, f'REF:{n_ref} SMP:{n_sample}'
    step = min(batch_size or n_sample, 300)
    return torch.cat([emd_approx(sample_pcs[s:s + step], ref_pcs[s:s + step]) for s in range(0, n_sample, step)])


def dist_chamfer(x, y, chamfer_loss=None):
    """utils/metrics.py:78-83: (for every point of y its squared distance to the nearest point of x,
    for every point of x the same towards y).  `chamfer_loss` is accepted for signature parity and unused: the
    distances come from the NN kernel, not from its (B,Nx,Ny) matrix (they agree to ~1e-7)."""
    d_x, d_y = nn_distance(x.contiguous(), y.contiguous())
    return d_y, d_x


def EMD_CD(sample_pcs, ref_pcs, batch_size, reduced=True, chamfer_loss=None):
    """utils/metrics.py:86-118 (the reference calls dist_chamfer without its third argument there and would raise;
    the evident intent — element-wise CD and EMD of matching clouds — is what this computes)."""
    n_sample, n_ref = sample_pcs.shape[0], ref_pcs.shape[0]
    assert n_sample == n_ref, f'REF:{n_ref} SMP:{n_sample}'
    cd_lst, emd_lst = [], []
    for s in range(0, n_sample, batch_size):
        a, b = sample_pcs[s:s + batch_size].contiguous(), ref_pcs[s:s + batch_size].contiguous()
        dl, dr = dist_chamfer(a, b, chamfer_loss)
        cd_lst.append(dl.mean(dim=1) + dr.mean(dim=1))
        emd_lst.append(emd_approx(a, b))
    cd, emd = torch.cat(cd_lst), torch.cat(emd_lst)
    if reduced:
        cd, emd = cd.mean(), emd.mean()
    return {'MMD-CD': cd, 'MMD-EMD': emd}


def _pairwise_EMD_CD_(sample_pcs, ref_pcs, batch_size, chamfer_loss=None):
    """utils/metrics.py:121-158 — all (sample, ref) pairs: returns (N_sample, N_ref) CD and EMD matrices.  One kernel
    batch per (sample, ref chunk): the sample cloud is expanded against up to `batch_size` reference clouds."""
    n_sample, n_ref = sample_pcs.shape[0], ref_pcs.shape[0]
    all_cd, all_emd = [], []
    for i in range(n_sample):
        cd_row, emd_row = [], []
        for r in range(0, n_ref, batch_size):
            ref_batch = ref_pcs[r:r + batch_size].contiguous()
            sample_exp = sample_pcs[i].view(1, -1, 3).expand(ref_batch.size(0), -1, -1).contiguous()
            dl, dr = dist_chamfer(sample_exp, ref_batch, chamfer_loss)
            cd_row.append((dl.mean(dim=1) + dr.mean(dim=1)).view(1, -1))
            emd_row.append(emd_approx(sample_exp, ref_batch).view(1, -1))
        all_cd.append(torch.cat(cd_row, dim=1))
        all_emd.append(torch.cat(emd_row, dim=1))
    return torch.cat(all_cd, dim=0), torch.cat(all_emd, dim=0)


def knn(Mxx, Mxy, Myy, k, sqrt=False):
    """utils/metrics.py:162-191 — leave-one-out k-NN two-sample test on precomputed distance matrices."""
    n0, n1 = Mxx.size(0), Myy.size(0)
    label = torch.cat((torch.ones(n0), torch.zeros(n1))).to(Mxx)
    M = torch.cat((torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.t(), Myy), 1)), 0)
    if sqrt:
        M = M.abs().sqrt()
    M = M + torch.diag(torch.full((n0 + n1,), float('inf')).to(Mxx))
    _, idx = M.topk(k, 0, False)
    votes = torch.zeros(n0 + n1).to(Mxx)
    for i in range(k):
        votes = votes + label.index_select(0, idx[i])
    pred = (votes >= float(k) / 2).float()
    s = {'tp': (pred * label).sum(), 'fp': (pred * (1 - label)).sum(),
         'fn': ((1 - pred) * label).sum(), 'tn': ((1 - pred) * (1 - label)).sum()}
    s.update({
        'precision': s['tp'] / (s['tp'] + s['fp'] + 1e-10),
        'recall': s['tp'] / (s['tp'] + s['fn'] + 1e-10),
        'acc_t': s['tp'] / (s['tp'] + s['fn'] + 1e-10),
        'acc_f': s['tn'] / (s['tn'] + s['fp'] + 1e-10),
        'acc': torch.eq(label, pred).float().mean(),
    })
    return s


def mmd_cov(all_dist):
    """utils/metrics.py:194-206 on an (N_sample, N_ref) distance matrix."""
    n_ref = all_dist.size(1)
    min_from_sample, min_idx = torch.min(all_dist, dim=1)
    min_to_ref, _ = torch.min(all_dist, dim=0)
    cov = torch.tensor(float(min_idx.unique().numel()) / float(n_ref)).to(all_dist)
    return {'mmd(Fidelity)': min_to_ref.mean(), 'cov(Coverage)': cov, 'mmd_smp': min_from_sample.mean()}


def compute_all_metrics(sample_pcs, ref_pcs, batch_size, chamfer_loss=None):
    """utils/metrics.py:209-238 (the 1-NN block is commented out in the reference as well)."""
    results = {}
    M_rs_cd, M_rs_emd = _pairwise_EMD_CD_(ref_pcs, sample_pcs, batch_size, chamfer_loss)
    results.update({"%s-CD" % k: v for k, v in mmd_cov(M_rs_cd.t()).items()})
    results.update({"%s-EMD" % k: v for k, v in mmd_cov(M_rs_emd.t()).items()})
    return results
