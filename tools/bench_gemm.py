#!/usr/bin/env python3
"""Micro-benchmark of the fp32-MFMA GEMM family on the step's shapes (GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
from hyperpocket_amd.ops import gemm  # noqa: E402


def timeit(fn, iters=100, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    shapes = [  # (M, N, K, trans_a, trans_b, ksplit, label)
        (65536, 512, 512, False, True, 1, "enc conv5 fwd"),
        (65536, 512, 256, False, True, 1, "enc conv4 fwd"),
        (65536, 256, 128, False, True, 1, "enc conv3 fwd"),
        (65536, 128, 64, False, True, 1, "enc conv2 fwd"),
        (32768, 256, 512, False, False, 1, "enc conv4 dX (Rc rows)"),
        (512, 256, 32768, True, False, 16, "enc conv4 dW (split-K 16)"),
        (64, 8320, 2048, False, True, 4, "hyper head fwd (M=64, split-K 4)"),
        (8320, 2048, 64, True, False, 1, "hyper head dW"),
        (64, 2048, 8320, False, False, 16, "hyper head dX (split-K 16)"),
        (4096, 4096, 4096, False, True, 1, "4096^3 reference point"),
    ]
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for M, N, K, ta, tb, ks, label in shapes:
        if only and only not in label:
            continue
        A = torch.randn((K, M) if ta else (M, K), device="cuda")
        B = torch.randn((N, K) if tb else (K, N), device="cuda")
        ms = timeit(lambda: gemm(A, B, trans_a=ta, trans_b=tb, ksplit=ks))
        print(f"{label:36s} M={M:6d} N={N:5d} K={K:6d}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
