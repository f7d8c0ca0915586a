"""Cull share of the EMD level sweeps by ORDERING of the two sets and tile geometry (CPU, numpy).
Extends tools/emd_cull_share.py (Morton order only): k-d order = recursive median split on the widest axis, which makes
every aligned run of 2^j consecutive points an axis-aligned box of exactly 2^j points.
share = fraction of (row tile x candidate block) tiles whose bounding boxes are further apart than the level's underflow radius.
"""
import sys
import numpy as np
sys.path.insert(0, "tools")
from emd_cull_share import morton_order, regimes, LEVELS, LOG2E


def kd_order(p, leaf=8):
    idx = np.arange(len(p))
    def rec(ix):
        if len(ix) <= leaf:
            return [ix]
        q = p[ix]
        ax = np.argmax(q.max(0) - q.min(0))
        o = ix[np.argsort(q[:, ax], kind="stable")]
        h = len(o) // 2
        return rec(o[:h]) + rec(o[h:])
    return np.concatenate(rec(idx))


def shares(rows_pts, cand_pts, order, rt, cb):
    a = rows_pts[order(rows_pts)].astype(np.float64)
    c = cand_pts[order(cand_pts)].astype(np.float64)
    nr, nc = len(a) // rt, len(c) // cb
    ab = a[:nr * rt].reshape(nr, rt, 3)
    cbx = c[:nc * cb].reshape(nc, cb, 3)
    gap = np.maximum(0.0, np.maximum(cbx.min(1)[None] - ab.max(1)[:, None], ab.min(1)[:, None] - cbx.max(1)[None]))
    dbox = (gap ** 2).sum(-1)
    d = ((a[:, None, :] - c[None, :, :]) ** 2).sum(-1)
    out = []
    for lv in LEVELS[:4]:
        thr = 150.0 / (abs(lv) * LOG2E)
        out.append(((dbox > thr).mean(), (d > thr).mean()))
    return out


if __name__ == "__main__":
    for name, (gt, rec) in regimes(per=4).items():
        for oname, order in (("morton", morton_order), ("kd", kd_order)):
            for rt in (16, 32, 64, 128):
                for cb in (8, 16):
                    acc = np.zeros((4, 2))
                    for g, r_ in zip(gt, rec):
                        for x, y in ((g, r_), (r_, g)):
                            acc += np.array(shares(x, y, order, rt, cb))
                    acc /= 2 * len(gt)
                    print(f"{name[:12]:12s} {oname:6s} rows {rt:3d} cands {cb:2d}  bbox: " + " ".join(f"{v:5.3f}" for v in acc[:, 0]) +
                          "  | pair-level: " + " ".join(f"{v:5.3f}" for v in acc[:, 1]), flush=True)
