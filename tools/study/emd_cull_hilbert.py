"""Cull shares with HILBERT order (one sort) against k-d order (eight segment sorts) and Morton: tools/study/emd_cull_order.py's measure."""
import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, "tools/study")
from emd_cull_share import morton_order, regimes, LEVELS, LOG2E
from emd_cull_order import kd_order, shares


def hilbert_index(q, bits):
    """Skilling's AxesToTranspose on integer coordinates q (n,3) -> Hilbert index (n,)."""
    x = q.astype(np.int64).T.copy()          # (3,n)
    n = 3
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(n):
            m = (x[i] & Q) != 0
            x[0] = np.where(m, x[0] ^ P, x[0])            # invert
            t = np.where(~m, (x[0] ^ x[i]) & P, 0)         # exchange
            x[0] ^= t
            x[i] ^= t
        Q >>= 1
    for i in range(1, n):
        x[i] ^= x[i - 1]
    t = np.zeros_like(x[0])
    Q = M
    while Q > 1:
        t = np.where((x[n - 1] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(n):
        x[i] ^= t
    # interleave: bit b of x[i] -> position 3*b + (2-i)
    h = np.zeros_like(x[0])
    for b in range(bits):
        for i in range(n):
            h |= ((x[i] >> b) & 1) << (3 * b + (n - 1 - i))
    return h


def hilbert_order(p, bits=7):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-12) * ((1 << bits) - 1)).astype(np.int64), 0, (1 << bits) - 1)
    return np.argsort(hilbert_index(q, bits), kind="stable")


def hilbert_rank_order(p, bits=7):
    """Hilbert order on per-axis RANK coordinates (adapts the grid to the marginal densities)."""
    n = len(p)
    r = np.empty((n, 3), np.int64)
    for a in range(3):
        r[np.argsort(p[:, a], kind="stable"), a] = np.arange(n)
    q = (r * (1 << bits)) // n
    return np.argsort(hilbert_index(q, bits), kind="stable")


if __name__ == "__main__":
    for name, (gt, rec) in regimes(per=4).items():
        for oname, order in (("morton", morton_order), ("hilbert7", hilbert_order), ("hilbert10", lambda p: hilbert_order(p, 10)),
                             ("hilb-rank", hilbert_rank_order), ("kd", kd_order)):
            for rt, cb in ((64, 8), (64, 16)):
                acc = np.zeros((4, 2))
                for g, r_ in zip(gt, rec):
                    for x, y in ((g, r_), (r_, g)):
                        acc += np.array(shares(x, y, order, rt, cb))
                acc /= 2 * len(gt)
                print(f"{name[:12]:12s} {oname:9s} rows {rt:3d} cands {cb:2d}  bbox: " + " ".join(f"{v:5.3f}" for v in acc[:, 0]), flush=True)
