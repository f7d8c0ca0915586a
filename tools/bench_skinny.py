#!/usr/bin/env python3
"""Split-K sweep for the hypernetwork's skinny (M = B = 64) GEMMs (GPU box only)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.ops import gemm
B, H, T = 64, 19011, 2048
x = torch.randn(B, T, device="cuda"); W = torch.randn(H, T, device="cuda") * 0.02; b = torch.zeros(H, device="cuda")
dth = torch.randn(B, H, device="cuda")
def t(fn): return bench.event_time_ms(fn, iters=20, warm=3) * 1e3
mb = H * T * 4 / 1e6
for ks in (1, 2, 3, 4, 6, 8, 12, 16):
    o = torch.empty(B, H, device="cuda")
    us = t(lambda: gemm(x, W, bias=b, ksplit=ks, out=o))
    print(f"heads fwd  ks={ks:3d} {us:7.1f} us  {mb/us:6.2f} TB/s(W)")
for ks in (8, 16, 32, 48, 64, 96, 128):
    o = torch.empty(B, T, device="cuda")
    us = t(lambda: gemm(dth, W, trans_b=False, ksplit=ks, out=o))
    print(f"heads dX   ks={ks:3d} {us:7.1f} us  {mb/us:6.2f} TB/s(W)")
o = torch.empty(H, T, device="cuda")
us = t(lambda: gemm(dth, x, trans_a=True, trans_b=False, out=o))
print(f"heads dW   ks=  1 {us:7.1f} us  {mb/us:6.2f} TB/s(dW)")
