"""Reference point for the flagship GEMM shape (encoder conv5: M=65536, N=K=512, fp32): what does the vendor library reach
through torch.matmul (hipBLASLt / rocBLAS, fp32, TF32 off)?  Not used by the product — a yardstick for gemm.hip."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
from hyperpocket_amd.ops import gemm
torch.backends.cuda.matmul.allow_tf32 = False
def timed(fn, iters=100, warm=200):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for M, N, K in ((65536, 512, 512), (65536, 512, 256), (65536, 256, 128), (8192, 8192, 8192)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    t_lib = timed(lambda: torch.matmul(A, W.t(), out=out))
    t_our = timed(lambda: gemm(A, W, out=out))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: library {t_lib*1e3:7.1f} us {fl/t_lib/1e9:6.1f} TF | gemm.hip {t_our*1e3:7.1f} us {fl/t_our/1e9:6.1f} TF")
