#!/usr/bin/env python3
"""Kernel-level timing of the fused target-network kernels via HIP events around direct C-ABI calls (GPU box only)."""
import os, sys, torch, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd._lib import call, current_stream, load_library
from hyperpocket_amd import ops
B, N = 64, 2048
theta = torch.randn(B, 19011, device="cuda") * 0.1
pts = ops.sample_points(B, N, 0.0, 1, 1, "cuda")
gy = torch.randn(B, N, 3, device="cuda")
y = torch.empty(B, N, 3, device="cuda"); gth = torch.empty_like(theta)
lib = load_library(); lib.hp_target_fused_workspace_floats.restype = ctypes.c_long
ws = torch.empty(lib.hp_target_fused_workspace_floats(B, N), device="cuda")
st = current_stream(theta.device)
f = bench.event_time_ms(lambda: call("hp_target_fused_forward", B, N, theta, 19011, pts, y, st), iters=50, warm=5) * 1e3
b = bench.event_time_ms(lambda: call("hp_target_fused_backward", B, N, theta, 19011, pts, gy, gth, ws, st), iters=50, warm=5) * 1e3
print(f"fused fwd {f:7.1f} us ({4.9e3 * B / 64 / f:5.1f} TF)   fused bwd(+reduce) {b:7.1f} us")
