"""GPU box: the P-format conv kernel (csrc/conv_pp.hip) against round 3's split kernel on the encoder's layer shapes (one encoder)."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-point-clouds-autocomplete_amd"))
import torch
from hyperpocket_amd import _lib, ops
lib = _lib.load_library()
lib.hp_gemm_pp_workspace_floats.restype = ctypes.c_long
f32 = dict(dtype=torch.float32, device="cuda")
M = int(os.environ.get("M", 65536))
st = _lib.current_stream(torch.device("cuda"))

def timeit(fn, n=50, warm=100):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

shapes = ((512, 512, 1, 256), (512, 512, 0, 256), (512, 256, 0, 256), (256, 128, 0, 128), (128, 64, 0, 64))
if os.environ.get('ONLY'):
    shapes = shapes[2:4] if os.environ.get('ONLY') == '2' else shapes[:1] + shapes[2:3]
for (N, K, mode, xcb) in shapes:
    X = torch.rand(M, K, **f32) ; W = torch.randn(N, K, **f32) * 0.05; b = torch.randn(N, **f32) * 0.01
    ws = torch.empty((lib.hp_gemm_pp_workspace_floats(ctypes.c_long(M), N, K),), **f32)
    _lib.call("hp_gemm_pp_prepare", ctypes.c_long(M), N, K, xcb, X, W, ws, st)
    t_pp = timeit(lambda: _lib.call("hp_gemm_pp_run", ctypes.c_long(M), N, K, xcb, b, 1, mode, 1024, ws, st))
    t_old = 1.0
    if not os.environ.get('ONLY'):
        old = ops.GemmF16x2(X, W, b, relu=True)
        t_old = timeit(old.run)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} mode={mode}: pp {t_pp:8.1f} us = {fl / t_pp / 1e6:7.1f} TFLOP/s alg ({3 * fl / t_pp / 1e6:7.1f} executed f16, "
          f"{fl / t_pp / 1e6 / 838.9:.3f} of f16/3) | round-3 kernel (store) {t_old:8.1f} us = {fl / t_old / 1e6:7.1f}", flush=True)
