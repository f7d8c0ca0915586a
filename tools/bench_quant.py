#!/usr/bin/env python3
"""GEMM rate vs number of 128x128 tiles (wave quantization of the conv5-shaped launch)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd.ops import gemm
w = torch.randn(512, 512, device="cuda") * 0.05; b = torch.zeros(512, device="cuda")
for tiles_m in (96, 128, 192, 256, 384, 448, 512, 576, 640, 768):
    m = tiles_m * 128
    a = torch.randn(m, 512, device="cuda"); c = torch.empty(m, 512, device="cuda")
    ms = bench.event_time_ms(lambda: gemm(a, w, bias=b, out=c), iters=20, warm=3)
    wgs = tiles_m * 4
    print(f"M={m:6d} WGs={wgs:5d} ({wgs/768:.2f} rounds of 768)  {ms*1e3:7.1f} us  {2.0*m*512*512/ms/1e9:6.1f} TFLOP/s")
