#!/usr/bin/env python3
"""conv_fwd.hip (X . W^T with W^T streamed from L2) against gemm.hip on the encoder's conv shapes (GPU box only)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-point-clouds-autocomplete_amd"))
import bench
from hyperpocket_amd import ops
from hyperpocket_amd._lib import call, current_stream, load_library

lib = load_library()


def desc(A, W, b, C, relu):
    d = ops._GemmDesc()
    M, K = A.shape
    N = W.shape[0]
    d.A, d.B, d.C, d.bias = A.data_ptr(), W.data_ptr(), C.data_ptr(), b.data_ptr()
    d.sAi, d.sAk, d.sBk, d.sBj, d.ldc = K, 1, 1, K, N
    d.M, d.N, d.K, d.batch, d.ksplit = M, N, K, 1, 1
    d.flags = 1 | (2 if relu else 0)
    return d


for (M, N, K) in [(65536, 512, 512), (65536, 512, 256), (65536, 256, 128), (65536, 128, 64), (131072, 512, 512)]:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    C1, C2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    Wt = torch.empty(K, N, device="cuda")
    d1, d2 = desc(A, W, b, C1, True), desc(A, W, b, C2, True)
    st = current_stream(A.device)
    assert lib.hp_conv_xwt_ok(ctypes.byref(d2)) == 1
    call("hp_conv_transpose_weights", 1, N, K, W, ctypes.c_long(0), Wt, ctypes.c_long(0), st)
    f_old = lambda: call("hp_gemm_f32", ctypes.byref(d1), st)
    f_new = lambda: call("hp_conv_xwt", ctypes.byref(d2), Wt, ctypes.c_long(0), st)
    f_old(); f_new()
    torch.cuda.synchronize()
    same = torch.equal(C1, C2)
    flops = 2.0 * M * N * K
    t_old = bench.event_time_ms(f_old, iters=50, warm=100)
    t_new = bench.event_time_ms(f_new, iters=50, warm=100)
    print(f"M={M} N={N} K={K}: gemm.hip {t_old*1e3:7.1f} us {flops/t_old/1e9:6.1f} TFLOP/s | xwt {t_new*1e3:7.1f} us {flops/t_new/1e9:6.1f} TFLOP/s | "
          f"bit-identical {same} (max diff {(C1-C2).abs().max().item():.2e})")
