"""VERDICT r5 item 2 — can the EMD sweeps take d^2 from the matrix pipe?  CPU feasibility, numpy fp32.

An MFMA can only form d^2 in the EXPANDED form  |p-c|^2 + |q-c|^2 - 2 (p-c).(q-c)  (c = a tile centre; K = 5: three products,
two norms).  The kernels (and approxmatch.cu:85,131,185) use the DIRECT form (p-q).(p-q), whose rounding error is RELATIVE to
d^2; the expanded form's is relative to |p-c|^2 + |q-c|^2 — absolute in d^2 — and the level multiplies it: exp(level * d^2) at
level -16384 turns an absolute 1e-8 into a relative 1.6e-4 of the exponential.

This script runs the nine-level algorithm in numpy fp32 with a pluggable distance and reports, against the direct
fp32 evaluation and against the fp64 evaluation of the C oracle:
  cost relative difference, grad2 max abs difference, fraction of grad2 components beyond the test bar (5e-5 + 1e-3 |x|).
Variants: direct | expanded, c = 0 | expanded, c = centre of the row tile (32 rows of the k-d ordered set: the tightest centring a
32x32 MFMA tile allows) | expanded, c = centre of the row tile, |.|^2 terms and the final sum in fp64 (only the dot product fp32:
what a split-precision epilogue could at best recover).
Usage: python tools/study/emd_expanded_distance.py [clouds-per-regime]   (N = 2048; ~1 min per cloud and variant)
"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "tools"); sys.path.insert(0, "tools/study")
from conftest import OracleLib
from emd_cull_share import regimes
from emd_cull_order import kd_order

LEVELS = [-(4.0 ** j) for j in range(7, -2, -1)]
f32 = np.float32


def dist_direct(p, q):
    d = (p[:, None, 0] - q[None, :, 0]) ** 2
    d += (p[:, None, 1] - q[None, :, 1]) ** 2
    d += (p[:, None, 2] - q[None, :, 2]) ** 2
    return d


def dist_expanded(p, q, centre="none", hi_norms=False):
    n = len(p)
    d = np.empty((n, len(q)), f32)
    T = 32
    for t0 in range(0, n, T):
        pt = p[t0:t0 + T]
        c = np.zeros(3, f32) if centre == "none" else ((pt.min(0) + pt.max(0)) * f32(0.5)).astype(f32)
        a, b = (pt - c).astype(f32), (q - c).astype(f32)
        dot = a[:, None, 0] * b[None, :, 0]
        dot = dot + a[:, None, 1] * b[None, :, 1]
        dot = dot + a[:, None, 2] * b[None, :, 2]          # fp32 accumulate, k order (the MFMA's)
        if hi_norms:
            na = (a.astype(np.float64) ** 2).sum(1)
            nb = (b.astype(np.float64) ** 2).sum(1)
            d[t0:t0 + T] = np.maximum(na[:, None] + nb[None, :] - 2.0 * dot.astype(np.float64), 0.0).astype(f32)
        else:
            na = (a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1] + a[:, 2] * a[:, 2]).astype(f32)
            nb = (b[:, 0] * b[:, 0] + b[:, 1] * b[:, 1] + b[:, 2] * b[:, 2]).astype(f32)
            d[t0:t0 + T] = np.maximum((na[:, None] + nb[None, :]) - f32(2) * dot, f32(0))
    return d


def emd(p, q, D):
    """p (n,3) = set1 rows k, q (m,3) = set2 cols l, D (n,m) fp32 squared distances.  -> cost, grad2 (m,3)."""
    n, m = D.shape
    remL, remR = np.ones(n, f32), np.ones(m, f32)
    M = np.zeros((n, m), f32)
    for lv in LEVELS:
        E = np.exp(f32(lv) * D, dtype=f32)
        ratioL = remL / (f32(1e-9) + E @ remR)
        sumr = (E.T @ ratioL) * remR
        ratioR = np.minimum(remR / (sumr + f32(1e-9)), f32(1)) * remR
        remR = np.maximum(f32(0), remR - sumr)
        W = (E * ratioL[:, None]) * ratioR[None, :]
        M += W
        remL = np.maximum(f32(0), remL - W.sum(1, dtype=f32))
    r = np.sqrt(D)
    cost = float((M.astype(np.float64) * r).sum())
    w = M / np.maximum(r, f32(1e-10))
    g2 = np.stack([(w * (q[None, :, a] - p[:, None, a])).sum(0, dtype=np.float64) for a in range(3)], 1)
    return cost, g2


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    o = OracleLib()
    variants = [("direct (shipped form)", lambda p, q: dist_direct(p, q)),
                ("expanded, c = 0", lambda p, q: dist_expanded(p, q, "none")),
                ("expanded, c = row-tile centre", lambda p, q: dist_expanded(p, q, "tile")),
                ("expanded, tile centre, fp64 norms", lambda p, q: dist_expanded(p, q, "tile", True))]
    print(f"{'regime':28s} {'variant':36s} cost rel vs direct | vs fp64    grad2 max abs vs direct   frac beyond bar")
    for name, (gt, rec) in regimes(per=per).items():
        for g, r_ in zip(gt, rec):
            p, q = g[kd_order(g)], r_[kd_order(r_)]
            _, c64 = o.approxmatch_f64(p[None], q[None])
            base = None
            for vname, fn in variants:
                cost, g2 = emd(p, q, fn(p, q))
                if base is None:
                    base = (cost, g2)
                err = np.abs(g2 - base[1])
                bad = (err > 5e-5 + 1e-3 * np.abs(base[1])).mean()
                print(f"{name:28s} {vname:36s} {abs(cost - base[0]) / base[0]:9.2e} | {abs(cost - c64[0]) / c64[0]:9.2e}   {err.max():9.2e}   {bad:9.2e}", flush=True)


if __name__ == "__main__":
    main()
