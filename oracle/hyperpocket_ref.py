"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

From-scratch PyTorch-CPU (fp32) restatement of the reference HyperPocket training step,
operating on a plain ``dict`` of parameter tensors keyed by the reference's state_dict
names.  Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.

Pinning: PINNED against the committed fixtures tests/golden/model_*.npz / train_steps.npz
/ points.npz / chamfer_*.npz, which were produced by importing the reference itself
(tests/golden/make_golden.py).  tests/test_oracle_golden.py is the check.

Reference files restated (cited per function):
  model/encoder.py, model/hyper_network.py, model/target_network.py,
  model/full_model.py, utils/points.py, losses/champfer_loss.py,
  core/epoch_loops.py:8-46, core/setup.py:12-19,63-77, core/main.py:62-66.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

TN_CHANNELS = (32, 64, 128, 64)          # settings/*.json.sample  target_network.layer_out_channels
ENC_CHANNELS = (3, 64, 128, 256, 512, 512)   # model/encoder.py:14-28
HN_TRUNK = (64, 128, 512, 1024, 2048)    # model/hyper_network.py:16-30


def target_layout(channels=TN_CHANNELS):
    """Offsets of (W,b) pairs inside the per-cloud weight vector (model/target_network.py:14-29,40-45).
    Returns list of (w_off, out, inp, b_off) and the total length (19011 for the default)."""
    dims = [3] + list(channels) + [3]
    off, out = 0, []
    for i in range(1, len(dims)):
        o, k = dims[i], dims[i - 1]
        out.append((off, o, k, off + o * k))
        off += o * k + o
    return out, off


def init_params(seed, random_out=128, real_out=128, channels=TN_CHANNELS):
    """Parameter dict with the reference's names, shapes and creation order, initialised as
    core/setup.py:63-77 does (xavier_uniform gain sqrt(2), zero bias).  Consumes the torch
    CPU RNG exactly like ``FullModel(cfg).apply(weights_init)`` after ``torch.manual_seed(seed)``
    (default nn init first, then xavier), so the same seed gives the same weights."""
    import torch.nn as nn
    torch.manual_seed(seed)
    mods = OrderedDict()

    def encoder(prefix, out_size):     # model/encoder.py:14-36
        for i, li in enumerate((0, 2, 4, 6, 8)):
            mods[f"{prefix}.conv.{li}"] = nn.Conv1d(ENC_CHANNELS[i], ENC_CHANNELS[i + 1], 1)
        mods[f"{prefix}.fc.0"] = nn.Linear(512, 512)
        mods[f"{prefix}.mu_layer"] = nn.Linear(512, out_size)
        mods[f"{prefix}.std_layer"] = nn.Linear(512, out_size)

    # model/full_model.py:28-41 construction order
    if random_out > 0:
        encoder("random_encoder", random_out)
    if real_out > 0:
        encoder("real_encoder", real_out)
    dims = (random_out + real_out,) + HN_TRUNK    # model/hyper_network.py:16-30
    for i, li in enumerate((0, 2, 4, 6, 8)):
        mods[f"hyper_network.model.{li}"] = nn.Linear(dims[i], dims[i + 1])
    tn = [3] + list(channels) + [3]
    for x in range(1, len(tn)):                   # model/hyper_network.py:32-36
        mods[f"hyper_network.output.{x - 1}"] = nn.Linear(2048, (tn[x - 1] + 1) * tn[x])
    gain = math.sqrt(2.0)
    # .apply(weights_init) visits modules in registration order (children before parents)
    for m in mods.values():
        torch.nn.init.xavier_uniform_(m.weight, gain)
        torch.nn.init.constant_(m.bias, 0)
    params = OrderedDict()
    for k, m in mods.items():
        params[k + ".weight"] = m.weight.detach().clone()
        params[k + ".bias"] = m.bias.detach().clone()
    return params


def generate_points(epoch, n, normalize=True, max_epoch=100):
    """utils/points.py:8-36 with the torch global CPU RNG (bit-exact draws)."""
    while True:
        pts = torch.zeros([n * 3, 3]).uniform_(-1, 1)
        pts = pts[torch.norm(pts, dim=1) < 1]
        if pts.shape[0] >= n:
            pts = pts[:n]
            break
    if normalize:
        coef = np.linspace(0, 1, max_epoch)[epoch - 1] if epoch <= max_epoch else 1
        nrm = np.linalg.norm(pts, axis=1)
        sel = nrm < coef
        if sel.any():
            sub = pts[sel]
            pts[sel] = coef * (sub.T / torch.from_numpy(np.linalg.norm(sub, axis=1)).float()).T
    return pts


def encoder_forward(P, prefix, x, is_vae, eps=None):
    """model/encoder.py:43-53; x is (B, N, 3) contiguous (the layout as loaded)."""
    h = x
    for i, li in enumerate((0, 2, 4, 6, 8)):
        w = P[f"{prefix}.conv.{li}.weight"]            # (Cout, Cin, 1)
        h = h @ w[:, :, 0].t() + P[f"{prefix}.conv.{li}.bias"]
        if i < 4:
            h = torch.relu(h)
    g = h.max(dim=1)[0]                                # max over points
    f = torch.relu(g @ P[f"{prefix}.fc.0.weight"].t() + P[f"{prefix}.fc.0.bias"])
    mu = f @ P[f"{prefix}.mu_layer.weight"].t() + P[f"{prefix}.mu_layer.bias"]
    if not is_vae:
        return mu
    lv = f @ P[f"{prefix}.std_layer.weight"].t() + P[f"{prefix}.std_layer.bias"]
    std = torch.exp(lv)                                # Q3: std = exp(logvar), not exp(0.5*)
    if eps is None:
        eps = torch.randn_like(std)
    return eps * std + mu, mu, torch.exp(lv)


def hypernet_forward(P, latent):
    """model/hyper_network.py:41-43"""
    h = latent
    for i, li in enumerate((0, 2, 4, 6, 8)):
        h = h @ P[f"hyper_network.model.{li}.weight"].t() + P[f"hyper_network.model.{li}.bias"]
        if i < 4:
            h = torch.relu(h)
    heads = []
    x = 0
    while f"hyper_network.output.{x}.weight" in P:
        heads.append(h @ P[f"hyper_network.output.{x}.weight"].t() + P[f"hyper_network.output.{x}.bias"])
        x += 1
    return torch.cat(heads, 1)


def target_forward(theta, pts, channels=TN_CHANNELS):
    """model/target_network.py:31-38 for one cloud: theta (19011,), pts (N,3) -> (N,3)"""
    layout, total = target_layout(channels)
    assert total == theta.numel()
    x = pts
    for li, (wo, o, k, bo) in enumerate(layout):
        x = torch.mm(x, theta[wo:wo + o * k].view(o, k).t()) + theta[bo:bo + o]
        if li < len(layout) - 1:
            x = torch.relu(x)
    return x


def chamfer_loss(preds, gts):
    """losses/champfer_loss.py:11-35 (expanded |x|^2+|y|^2-2xy form, batch SUM)."""
    x, y = gts, preds
    xx = torch.bmm(x, x.transpose(2, 1))
    yy = torch.bmm(y, y.transpose(2, 1))
    zz = torch.bmm(x, y.transpose(2, 1))
    rx = torch.diagonal(xx, dim1=1, dim2=2).unsqueeze(1).expand_as(zz.transpose(2, 1))
    ry = torch.diagonal(yy, dim1=1, dim2=2).unsqueeze(1).expand_as(zz)
    Pm = rx.transpose(2, 1) + ry - 2 * zz
    return torch.min(Pm, 1)[0].sum() + torch.min(Pm, 2)[0].sum()


class _OracleEMD(torch.autograd.Function):
    """match_cost (utils/pytorch_structural_losses/match_cost.py:5-48) on CPU through the C restatement
    (oracle/libstructural_losses_ref.so): forward = ApproxMatch + MatchCost, backward = MatchCostGrad * grad."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes
            import os
            import subprocess
            here = os.path.dirname(os.path.abspath(__file__))
            so = os.path.join(here, "libstructural_losses_ref.so")
            if not os.path.exists(so):
                subprocess.check_call(["make", "-C", here])
            cls._lib = ctypes.CDLL(so)
        return cls._lib

    @staticmethod
    def forward(ctx, a, b):
        import ctypes
        a, b = a.detach().contiguous().float(), b.detach().contiguous().float()
        B, n, m = a.shape[0], a.shape[1], b.shape[1]
        match, temp, out = torch.empty(B, m, n), torch.empty(B, 2 * (n + m)), torch.empty(B)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        lib = _OracleEMD.lib()
        lib.ref_approxmatch(B, n, m, P(a), P(b), P(match), P(temp))
        lib.ref_matchcost(B, n, m, P(a), P(b), P(match), P(out))
        ctx.save_for_backward(a, b, match)
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        a, b, match = ctx.saved_tensors
        B, n, m = a.shape[0], a.shape[1], b.shape[1]
        g1, g2 = torch.empty(B, n, 3), torch.empty(B, m, 3)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _OracleEMD.lib().ref_matchcostgrad(B, n, m, P(a), P(b), P(match), P(g1), P(g2))
        ge = g.view(-1, 1, 1)
        return g1 * ge, g2 * ge


def match_cost(a, b):
    return _OracleEMD.apply(a, b)


def mode_of(P):
    has_rand = "random_encoder.fc.0.weight" in P
    has_real = "real_encoder.fc.0.weight" in P
    return "HyperPocket" if has_rand and has_real else ("HyperCloud" if has_rand else "HyperRec")


def full_forward(P, existing, missing, points, eps=None, training=True, noise=None):
    """model/full_model.py:54-80,98-152.  existing/missing are (B,N,3); points (B,Ngt,3) are the
    decoder input samples (injected: the reference draws them per cloud on the CPU).  Returns
    (rec (B,3,Ngt), exp_logvar, mu, theta)."""
    mode = mode_of(P)
    mu = explv = None
    if mode == "HyperPocket":
        if training:
            z, mu, explv = encoder_forward(P, "random_encoder", missing, True, eps)
        elif noise is None:
            _, z, _ = encoder_forward(P, "random_encoder", missing, True, torch.zeros(missing.shape[0],
                                      P["random_encoder.mu_layer.bias"].numel()))
        else:
            z = noise
        latent = torch.cat([z, encoder_forward(P, "real_encoder", existing, False)], 1)
        if not training:
            mu = explv = None
    elif mode == "HyperRec":
        latent = encoder_forward(P, "real_encoder", existing, False)
    else:
        if training:
            latent, mu, explv = encoder_forward(P, "random_encoder", existing, True, eps)
        elif noise is None:
            _, latent, _ = encoder_forward(P, "random_encoder", existing, True,
                                           torch.zeros(existing.shape[0], P["random_encoder.mu_layer.bias"].numel()))
        else:
            latent = noise
    theta = hypernet_forward(P, latent)
    rec = torch.stack([target_forward(theta[j], points[j]).t() for j in range(theta.shape[0])])
    return rec, explv, mu, theta


def step_loss(P, existing, missing, gt, points, eps, loss_coef=0.05, emd_coef=0.0):
    """core/epoch_loops.py:23-31 -> (loss_all, loss_r, loss_kld, rec).  emd_coef > 0 adds the bench's "Chamfer+EMD"
    term emd_coef * sum_b match_cost(gt, rec)_b / N (utils/metrics.py:71-76 normalisation)."""
    rec, explv, mu, _ = full_forward(P, existing, missing, points, eps, training=True)
    rec_n3 = rec.permute(0, 2, 1)
    loss_r = torch.mean(loss_coef * chamfer_loss(gt, rec_n3))
    loss_all = loss_r
    kld = None
    if mode_of(P) == "HyperPocket":
        kld = 0.5 * (torch.exp(explv) + torch.square(mu) - 1 - explv).sum() / existing.shape[0]
        loss_all = loss_all + kld
    if emd_coef:
        loss_all = loss_all + emd_coef * (match_cost(gt, rec_n3.contiguous()) / float(gt.shape[1])).sum()
    return loss_all, loss_r, kld, rec


class Adam:
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0, amsgrad=False) restated
    (core/main.py:62-66, settings/config.json.sample:12-29)."""

    def __init__(self, P, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, betas[0], betas[1], eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.seen = set()

    def step(self, P, grads):
        self.t += 1
        for k, g in grads.items():
            if g is None:       # parameters that never receive a gradient are skipped
                continue
            self.seen.add(k)
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            P[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def train_step(P, opt, existing, missing, gt, points, eps, loss_coef=0.05, emd_coef=0.0):
    """One reference training step (core/epoch_loops.py:15-39) on the parameter dict P (in place)."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    loss_all, loss_r, kld, rec = step_loss(leaves, existing, missing, gt, points, eps, loss_coef, emd_coef)
    loss_all.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    opt.step(P, grads)
    return loss_all.detach(), loss_r.detach(), (None if kld is None else kld.detach()), rec.detach(), grads
