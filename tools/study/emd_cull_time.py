"""hp_emd_forward at B=64, N=2048 in one regime, with hp_emd_set_cull(K): for rocprofv3 --kernel-trace --stats.
usage: emd_cull_time.py K [regime-index 0..2] [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "study")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from emd_cull_check import lib, timeit
from emd_cull_share import regimes
K = int(sys.argv[1]); ri = int(sys.argv[2]) if len(sys.argv) > 2 else 1; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
name, (gt, rec) = list(regimes(per=64).items())[ri]
lib.hp_emd_set_cull(K)
print(name, "cull", K, f"{timeit(gt, rec, iters):.4f} ms")
