"""GPU box: HP_PP_PROF=1 — s_memtime stamps of one k-tile of conv_pp_kernel (waves 0 / 4 of workgroup 0, second tile, middle k-tile)."""
import ctypes, os, sys
os.environ["HP_PP_PROF"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-point-clouds-autocomplete_amd"))
import torch
from hyperpocket_amd import _lib
lib = _lib.load_library()
lib.hp_gemm_pp_workspace_floats.restype = ctypes.c_long
f32 = dict(dtype=torch.float32, device="cuda")
st = _lib.current_stream(torch.device("cuda"))

for (M, N, K, mode, xcb) in ((65536, 512, 512, 1, 256), (65536, 512, 256, 0, 256)):
    X = torch.rand(M, K, **f32); W = torch.randn(N, K, **f32) * 0.05; b = torch.randn(N, **f32) * 0.01
    ws = torch.empty((lib.hp_gemm_pp_workspace_floats(ctypes.c_long(M), N, K),), **f32)
    _lib.call("hp_gemm_pp_prepare", ctypes.c_long(M), N, K, xcb, X, W, ws, st)
    for _ in range(50):
        _lib.call("hp_gemm_pp_run", ctypes.c_long(M), N, K, xcb, b, 1, mode, 1024, ws, st)
    out = (ctypes.c_ulonglong * 128)()
    assert lib.hp_conv_pp_prof(out) == 0
    for g in (0, 1):
        t = [out[g * 64 + i] for i in range(14)]
        print(f"K={K} mode={mode} group {g}: " + " ".join(f"{names[i]}={t[i] - t[i - 1]}" for i in range(1, 14)) + f" | k-tile {t[13] - t[0]} (100 MHz ticks x ~21 = cycles)")
