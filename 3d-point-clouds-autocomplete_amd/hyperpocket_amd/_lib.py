"""ctypes loader for libhyperpocket_hip.so — the only door between Python and the HIP kernels."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhyperpocket_hip.so")
_LIB = None


class HipExtensionError(RuntimeError):
    """Raised when the HIP library is missing/unloadable or a kernel launch fails.

    Mirrors the reference binding's failure mode (structural_loss.cpp:7-9 AT_ASSERTM ->
    RuntimeError; approxmatch.cu:334-337 std::runtime_error -> RuntimeError)."""


def library_path():
    return _SO


def load_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise HipExtensionError(
                f"{_SO} not found: build it with `python 3d-point-clouds-autocomplete_amd/build.py` "
                "(there is no CPU fallback for the HyperPocket hot path)")
        try:
            _LIB = ctypes.CDLL(_SO)
        except OSError as e:  # pragma: no cover
            raise HipExtensionError(f"cannot load {_SO}: {e}") from e
        for name in ("hp_approxmatch_workspace_floats", "hp_matchcost_workspace_floats",
                     "hp_chamfer_workspace_floats"):
            getattr(_LIB, name).restype = ctypes.c_long
    return _LIB


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def current_stream(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check_input(t, name, dtype=torch.float32):
    """structural_loss.cpp:7-9 CHECK_CUDA / CHECK_CONTIGUOUS (+ dtype, which the reference assumes)."""
    if not t.is_cuda:
        raise HipExtensionError(f"{name} must be a CUDA (HIP) tensor — the HyperPocket kernels have no CPU path")
    if not t.is_contiguous():
        raise HipExtensionError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise HipExtensionError(f"{name} must be {dtype}, got {t.dtype}")


def call(fn_name, *args):
    """Invoke an int-returning entry point; non-zero return -> HipExtensionError."""
    fn = getattr(load_library(), fn_name)
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            conv.append(ptr(a))
        elif a is None:
            conv.append(ctypes.c_void_p(0))
        elif isinstance(a, float):
            conv.append(ctypes.c_float(a))
        elif isinstance(a, bool):
            conv.append(ctypes.c_int(int(a)))
        elif isinstance(a, int):
            conv.append(ctypes.c_longlong(a) if abs(a) > 0x7FFFFFFF else ctypes.c_int(a))
        else:
            conv.append(a)
    rc = fn(*conv)
    if rc != 0:
        raise HipExtensionError(f"HIP kernel failed : {fn_name} returned {rc}")
