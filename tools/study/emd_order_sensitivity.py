"""Does re-ordering the two sets (k-d order, what a culling sweep needs) keep the EMD inside the parity bars?
CPU: the C oracle on the original order against the C oracle on k-d ordered inputs (results permuted back)."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "tools"); sys.path.insert(0, "tools/study")
from conftest import OracleLib
from emd_cull_share import regimes
from emd_cull_order import kd_order
o = OracleLib()
for name, (gt, rec) in regimes(per=4).items():
    om, _ = o.approxmatch(gt, rec); c0 = o.matchcost(gt, rec, om); _, g20 = o.matchcostgrad(gt, rec, om)
    pg = [kd_order(g) for g in gt]; pr = [kd_order(r) for r in rec]
    gts = np.stack([g[p] for g, p in zip(gt, pg)]); recs = np.stack([r[p] for r, p in zip(rec, pr)])
    om2, _ = o.approxmatch(gts, recs); c1 = o.matchcost(gts, recs, om2); _, g21 = o.matchcostgrad(gts, recs, om2)
    g2b = np.empty_like(g21)
    for i, p in enumerate(pr): g2b[i][p] = g21[i]
    err = np.abs(g2b - g20)
    bad = (err > 5e-5 + 1e-3 * np.abs(g20)).mean()
    print(f"{name:28s} cost rel diff {np.abs(c1-c0)/np.maximum(c0,1e-30)}  grad2 max err {err.max():.2e} frac beyond bar {bad:.2e}")
