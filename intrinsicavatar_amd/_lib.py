"""ctypes loader for libia_amd.so -- the ONLY compute backend of this package.

There is deliberately no CPU / PyTorch fallback: if the HIP library is missing or a
tensor is not on a GPU, the call fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libia_amd.so")
_lib = None


class IaError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise IaError(
                f"{_SO} not found: build it with `python -m intrinsicavatar_amd.build` "
                "(there is no CPU fallback for the MI355X hot path)")
        _lib = C.CDLL(_SO)
        _lib.ia_last_error.restype = C.c_char_p
        _lib.ia_scan_tmp_bytes.restype = C.c_int64
        _lib.ia_scan_tmp_bytes.argtypes = [C.c_int64]
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().ia_last_error().decode(errors="replace")
        raise IaError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise IaError("intrinsicavatar_amd operators need GPU tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise IaError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def scan_tmp(n: int, device, extra_bytes: int = 0):
    nbytes = int(lib().ia_scan_tmp_bytes(C.c_int64(max(int(n), 1)))) + int(extra_bytes) + 64
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


i64 = C.c_int64
i32 = C.c_int
f32 = C.c_float
