"""How long does a culling launch take when NOTHING survives (two sets far apart)?  The fixed cost of a launch: dispatch, row data,
boxes, ballot, LDS reduction, stores.  For rocprofv3 --kernel-trace --stats."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "study")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from emd_cull_check import lib, timeit
import numpy as np
r = np.random.RandomState(0)
gt = r.rand(64, 2048, 3).astype(np.float32) - 0.5
rec = r.rand(64, 2048, 3).astype(np.float32) - 0.5 + np.float32(5.0)
lib.hp_emd_set_cull(3)
print("far apart, cull 3:", timeit(gt, rec, 20))
