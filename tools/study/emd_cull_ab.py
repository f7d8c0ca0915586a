"""A/B of hp_emd_set_cull settings in ONE process (box-to-box noise is ~15 us): alternating rounds, B=64, N=2048."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "study")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from emd_cull_check import lib, timeit
from emd_cull_share import regimes
import numpy as np
Ks = [int(a) for a in sys.argv[1:]] or [0, 2, 3, 4]
for name, (gt, rec) in regimes(per=64).items():
    res = {k: [] for k in Ks}
    for rnd in range(5):
        for k in Ks:
            lib.hp_emd_set_cull(k)
            res[k].append(timeit(gt, rec, 20))
    print(f"{name:28s} " + "  ".join(f"cull={k}: {np.median(v):.4f} (min {min(v):.4f})" for k, v in res.items()), flush=True)
