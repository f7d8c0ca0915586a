"""hp_emd_forward(B=64, N=2048, grad2) call time for every rows-per-lane setting (GPU box)."""
import os, sys, itertools, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd._lib import call
torch.cuda.set_device(0)
for r1, r2, g2 in [(0, 0, 0)] + list(itertools.product((1, 2, 4), (1, 2, 4), (1, 2))):
    call("hp_emd_set_rows_per_lane", r1, r2, g2)
    ms = bench.roofline_emd(64, 2048)["avg_call_ms"]
    print(f"rows1={r1} rows2={r2} grad2={g2}: {ms:.4f} ms", flush=True)
