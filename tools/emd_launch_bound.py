#!/usr/bin/env python3
"""What could a persistent per-cloud EMD kernel save at most?  GPU box.

The 19 sweep launches of one hp_emd_forward call each pay a ramp (dispatch, prologue) and a tail (the last workgroups
finishing alone); a persistent kernel with free hand-offs between the phases would pay them once.  The same launches on
k x 64 clouds amortise those fixed costs over k times the work while every wave runs the identical instruction stream, so
    T(64) - T(64 k) / k
bounds from above what removing the launch boundaries of the B = 64 call can gain (a real persistent kernel also pays its
flag hand-offs: ~3-4 us each with agent-scope release / acquire, docs/DESIGN_HISTORY.md 7b)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

torch.cuda.set_device(0)
out = {"what": __doc__.split("\n\n")[1].replace("\n", " "), "n": 2048, "calls": []}
base = None
for k in (1, 2, 4, 9):
    ms = min(bench.roofline_emd(64 * k, 2048)["avg_call_ms"] for _ in range(3))
    base = base or ms
    out["calls"].append({"clouds": 64 * k, "ms_per_call": round(ms, 4), "ms_per_64_clouds": round(ms / k, 4),
                         "saving_bound_ms_at_B64": round(base - ms / k, 4)})
    print(out["calls"][-1], flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
