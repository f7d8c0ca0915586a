#!/usr/bin/env python3
"""Micro-benchmark of the structural-loss kernels at the BASELINE shapes (GPU box only)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
from hyperpocket_amd.utils.pytorch_structural_losses import StructuralLossesBackend as B  # noqa: E402
from hyperpocket_amd.losses.champfer_loss import ChamferLoss  # noqa: E402


def timeit(fn, iters=20, warm=3):
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
    for b, n in [(64, 2048), (32, 2048), (64, 8192)]:
        x = torch.rand(b, n, 3, device="cuda") - 0.5
        y = torch.rand(b, n, 3, device="cuda") - 0.5
        t = timeit(lambda: B.NNDistance(x, y))
        pairs = 2.0 * b * n * n
        print(f"NNDistance b={b} n={n}: {t:.3f} ms  {pairs / t / 1e6:.1f} Gpair/s  {pairs * 8 / t / 1e9:.2f} TFLOP/s(8/pair)")
        L = ChamferLoss()
        xr = x.clone().requires_grad_(True)

        def fb():
            xr.grad = None
            L(y, xr).backward()
        print(f"Chamfer fwd+bwd b={b} n={n}: {timeit(fb):.3f} ms")
        if n <= 2048:
            t = timeit(lambda: B.ApproxMatch(x, y), iters=5, warm=1)
            print(f"ApproxMatch b={b} n={n}: {t:.3f} ms   match write {b * n * n * 4 / t / 1e6:.1f} GB/s; "
                  f"{36.0 * b * n * n / t / 1e6:.1f} Gexp/s")
            match, _ = B.ApproxMatch(x, y)
            t = timeit(lambda: B.MatchCost(x, y, match), iters=10)
            print(f"MatchCost b={b} n={n}: {t:.3f} ms   {b * n * n * 4 / t / 1e6:.1f} GB/s")
            t = timeit(lambda: B.MatchCostGrad(x, y, match), iters=10)
            print(f"MatchCostGrad b={b} n={n}: {t:.3f} ms   {2 * b * n * n * 4 / t / 1e6:.1f} GB/s")
            del match
            from hyperpocket_amd.utils.pytorch_structural_losses.match_cost import match_cost
            yr = y.clone().requires_grad_(True)

            def emd_fb():
                yr.grad = None
                match_cost(x, yr).sum().backward()
            print(f"match_cost fused fwd+bwd b={b} n={n}: {timeit(emd_fb, iters=5, warm=1):.3f} ms")


if __name__ == "__main__":
    main()
