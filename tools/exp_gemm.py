#!/usr/bin/env python3
"""Times hp_gemm_f32 of experimental builds (tools/exp/*.so) on the encoder conv5 shape.

Used in round 1 to A/B kernel variants built with -D flags from csrc/gemm.hip into tools/exp/ (generic vs branch-free
loaders: 85 -> 113 TFLOP/s; BK 16 vs 32; s_setprio; no-load ceiling 122 TFLOP/s).  Build a variant with
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared csrc/gemm.hip -o tools/exp/gemm_x.so
"""
import ctypes, glob, os, sys
import torch
from ctypes import c_int, c_long, c_void_p

class Desc(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("mask", c_void_p), ("add", c_void_p), ("ws", c_void_p),
                ("sAz", c_long), ("sBz", c_long), ("sCz", c_long), ("sBiasz", c_long), ("sMaskz", c_long), ("sAddz", c_long),
                ("sAi", c_long), ("sAk", c_long), ("sBk", c_long), ("sBj", c_long),
                ("ldc", c_int), ("ldmask", c_int), ("ldadd", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("batch", c_int), ("ksplit", c_int), ("flags", c_int)]

def run(so, M, N, K):
    lib = ctypes.CDLL(so)
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    d = Desc(); d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.sAi, d.sAk, d.sBk, d.sBj, d.ldc, d.M, d.N, d.K, d.batch = K, 1, 1, K, N, M, N, K, 1
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib.hp_gemm_f32(ctypes.byref(d), st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    ref = (A[:256].double() @ B.double().t()).float()
    err = (C[:256] - ref).abs().max().item()
    return ms, 2.0 * M * N * K / ms / 1e9, err

for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "*.so"))):
    for (M, N, K) in [(65536, 512, 512), (4096, 4096, 4096)]:
        ms, tf, err = run(so, M, N, K)
        print(f"{os.path.basename(so):28s} M={M} N={N} K={K}: {ms*1e3:8.1f} us {tf:7.1f} TFLOP/s  maxerr {err:.2e}")
