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


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream(device=None):
    """The caller's current HIP stream as a raw handle.  (torch.cuda.current_stream builds a Stream object through
    several Python layers: ~7 us per call, a dozen calls per step; the raw getter is a single C call.)"""
    if _RAW_STREAM is not None:
        if device is None:
            idx = torch.cuda.current_device()
        elif type(device) is int:
            idx = device
        else:
            if isinstance(device, str):
                device = torch.device(device)
            idx = device.index if device.index is not None else torch.cuda.current_device()
        return ctypes.c_void_p(_RAW_STREAM(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check_input(t, name, dtype=torch.float32):
    """structural_loss.cpp:7-9 CHECK_CUDA / CHECK_CONTIGUOUS (+ dtype, which the reference assumes)."""
    if not t.is_cuda:
        raise HipExtensionError(f"{name} must be a CUDA (HIP) tensor — the HyperPocket kernels have no CPU path")
    if not t.is_contiguous():
        raise HipExtensionError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise HipExtensionError(f"{name} must be {dtype}, got {t.dtype}")


_FN = {}
_VOIDP, _FLOAT, _INT, _LONGLONG = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_longlong
_NULL = ctypes.c_void_p(0)


def call(fn_name, *args):
    """Invoke an int-returning entry point; non-zero return -> HipExtensionError."""
    fn = _FN.get(fn_name)
    if fn is None:
        fn = _FN[fn_name] = getattr(load_library(), fn_name)
    conv = []
    for a in args:
        t = type(a)
        if t is torch.Tensor or t is torch.nn.Parameter:
            conv.append(_VOIDP(a.data_ptr()))
        elif a is None:
            conv.append(_NULL)
        elif t is float:
            conv.append(_FLOAT(a))
        elif t is bool:
            conv.append(_INT(int(a)))
        elif t is int:
            conv.append(_LONGLONG(a) if abs(a) > 0x7FFFFFFF else _INT(a))
        elif isinstance(a, torch.Tensor):
            conv.append(_VOIDP(a.data_ptr()))
        else:
            conv.append(a)
    rc = fn(*conv)
    if rc != 0:
        raise HipExtensionError(f"HIP kernel failed : {fn_name} returned {rc}")
