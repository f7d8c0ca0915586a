#!/usr/bin/env python3
"""Both encoders' conv stacks (L2..L5 shapes, B=64 x 1024 points): two chains back to back, two chains on two streams,
or ONE batched (z = 2) launch per layer — does batching amortise the per-launch fixed cost?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
from hyperpocket_amd.ops import gemm
R = 65536
dims = [64, 128, 256, 512, 512]
X = torch.randn(2, R, 64, device="cuda")
W = [torch.randn(2, dims[i + 1], dims[i], device="cuda") * 0.05 for i in range(4)]
bias = [torch.randn(2, dims[i + 1], device="cuda") * 0.05 for i in range(4)]
H = [torch.empty(2, R, dims[i + 1], device="cuda") for i in range(4)]

def chain(z):
    a = X[z]
    for i in range(4):
        gemm(a, W[i][z], bias=bias[i][z], relu=True, out=H[i][z])
        a = H[i][z]

def batched():
    a = X
    for i in range(4):
        gemm(a, W[i], bias=bias[i], relu=True, out=H[i])
        a = H[i]

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): chain(0)
    with torch.cuda.stream(s2): chain(1)
    cur.wait_stream(s1); cur.wait_stream(s2)

def timed(fn, iters=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

flops = 2 * 2.0 * R * sum(dims[i] * dims[i + 1] for i in range(4))
for name, fn in (("sequential", lambda: (chain(0), chain(1))), ("two streams", two_streams), ("batched z=2", batched)):
    ms = timed(fn)
    print(f"{name:12s} {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
