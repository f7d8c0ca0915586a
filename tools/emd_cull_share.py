"""How many (row wave, candidate block) tiles of the EMD sweeps consist of exponentials that are all EXACTLY zero?
(measure before building.)  CPU only, numpy.

exp2(level*log2(e)*d) is +0 in fp32 once its argument is below -150 (2^-149 is the smallest denormal; v_exp_f32 flushes
already at -126), so a sweep may skip a tile — all its terms are exact zeros: fma(0, w, acc) == acc — when the smallest
squared distance between the tile's rows and its candidates exceeds 150 / (|level| * log2 e).  For that to happen often
the rows of a wave and the candidates of a block must be spatially compact: both sets are put in Morton (Z-curve) order
first.  Reported per level, for tiles of `rows` consecutive rows x 16 consecutive candidates:
  ideal  share of tiles whose true minimum distance exceeds the threshold
  bbox   share a kernel can decide from the two bounding boxes alone (what a scalar-path test would see)
Regimes: tests/test_structural_losses_gpu.py::_emd_regimes (uniform, noisy copy, clustered), N = 2048.
"""
import json
import sys

import numpy as np

LOG2E = 1.4426950408889634
LEVELS = [-16384.0, -4096.0, -1024.0, -256.0, -64.0, -16.0, -4.0, -1.0, -0.25]


def morton_order(p):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-12) * 1023).astype(np.int64), 0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return np.argsort(code, kind="stable")


def regimes(seed=20260, n=2048, per=6):
    r = np.random.RandomState(seed)
    out = {}
    gt = r.rand(per, n, 3).astype(np.float32) - 0.5
    out["uniform vs uniform"] = (gt, r.rand(per, n, 3).astype(np.float32) - 0.5)
    gt = r.rand(per, n, 3).astype(np.float32) - 0.5
    sig = np.resize(np.array([0.002, 0.01, 0.02, 0.05], np.float32), per)[:, None, None]
    perm = np.stack([g[r.permutation(n)] for g in gt])
    out["noisy copy (late training)"] = (gt, perm + sig * r.randn(per, n, 3).astype(np.float32))
    gts, recs = [], []
    for i in range(per):
        k = 3 + i % 10
        cen = (r.rand(k, 3) - 0.5) * 0.8
        s = 0.01 + 0.05 * r.rand(k, 1)
        za, zb = r.randint(0, k, n), r.randint(0, k, n)
        gts.append(cen[za] + s[za] * r.randn(n, 3))
        recs.append(cen[zb] + s[zb] * r.randn(n, 3))
    out["clustered"] = (np.asarray(gts, np.float32), np.asarray(recs, np.float32))
    return out


def tile_shares(rows_pts, cand_pts, rows_per_tile, cands_per_tile=16):
    a = rows_pts[morton_order(rows_pts)].astype(np.float64)
    c = cand_pts[morton_order(cand_pts)].astype(np.float64)
    d = ((a[:, None, :] - c[None, :, :]) ** 2).sum(-1)
    nr, nc = len(a) // rows_per_tile, len(c) // cands_per_tile
    dmin = d[:nr * rows_per_tile, :nc * cands_per_tile].reshape(nr, rows_per_tile, nc, cands_per_tile).min(axis=(1, 3))
    ab = a[:nr * rows_per_tile].reshape(nr, rows_per_tile, 3)
    cb = c[:nc * cands_per_tile].reshape(nc, cands_per_tile, 3)
    gap = np.maximum(0.0, np.maximum(cb.min(1)[None] - ab.max(1)[:, None], ab.min(1)[:, None] - cb.max(1)[None]))
    dbox = (gap ** 2).sum(-1)
    ideal, bbox = [], []
    for lv in LEVELS:
        thr = 150.0 / (abs(lv) * LOG2E)
        ideal.append(float((dmin > thr).mean()))
        bbox.append(float((dbox > thr).mean()))
    return ideal, bbox


def main():
    res = {"levels": LEVELS, "threshold_d2": [150.0 / (abs(lv) * LOG2E) for lv in LEVELS], "regimes": {}}
    for name, (gt, rec) in regimes().items():
        res["regimes"][name] = {}
        for rows in (64, 128, 256):
            acc_i, acc_b = np.zeros(len(LEVELS)), np.zeros(len(LEVELS))
            for g, r_ in zip(gt, rec):
                # rows = set1 (gt) against candidates = set2 (rec), and the transposed sweeps
                for x, y in ((g, r_), (r_, g)):
                    i, b = tile_shares(x, y, rows)
                    acc_i += i
                    acc_b += b
            k = 2 * len(gt)
            res["regimes"][name][f"rows{rows}"] = {"ideal": [round(v / k, 4) for v in acc_i], "bbox": [round(v / k, 4) for v in acc_b]}
            print(f"{name:28s} rows/tile {rows:3d}  bbox-decidable all-zero share per level:",
                  " ".join(f"{v / k:5.2f}" for v in acc_b[:5]), "| ideal:", " ".join(f"{v / k:5.2f}" for v in acc_i[:5]), flush=True)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
