"""ctypes loader for libia_amd.so -- the ONLY compute backend of this package.

There is deliberately no CPU / PyTorch fallback: if the HIP library is missing or a
tensor is not on a GPU, the call fails loudly.
"""
import collections
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libia_amd.so")
_lib = None


class IaError(RuntimeError):
    pass


# optional outputs that change an entry point's compulsory traffic: ia_fuse_broyden(..., x, J_inv, is_valid, fwd_J, stream)
_EXTRAS = {"ia_fuse_broyden": lambda a: dict(I=int(a[2].value), J_inv=bool(a[16].value), fwd_J=bool(a[18].value))}


class _Timed:
    """optional per-entry-point HIP-event timing (bench.py): events are recorded on the stream the
    kernels are launched on (torch's current stream), nothing synchronises until report()."""

    def __init__(self, cdll):
        self._cdll = cdll
        self.enabled = False
        self.events = {}

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if not name.startswith("ia_") or name in ("ia_last_error", "ia_scan_tmp_bytes", "ia_version", "ia_hashgrid_n_entries", "ia_traverse_scratch_bytes", "ia_occgrid_tmp_bytes", "ia_hashgrid_bwd_scratch_bytes", "ia_traverse_fused_scratch_bytes", "ia_hashgrid_fwd_scratch_bytes", "ia_eikonal_partials"):
            return fn

        def call(*args):
            if not self.enabled:
                return fn(*args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            # units of the launch = its first int64 argument (every entry point leads with its element count);
            # per-entry extras (which optional outputs were requested) for the algorithmic-bytes model of bench.py
            units = next((int(a.value) for a in args if isinstance(a, C.c_int64)), 0)
            ex = _EXTRAS.get(name)
            self.events.setdefault(name, []).append((e0, e1, units, ex(args) if ex else None))
            return rc
        return call

    def start(self):
        self.events = {}
        self.enabled = True

    def report(self, detail: bool = False):
        """-> {entry point: (n_calls, total_ms)}; synchronises.  detail=True: {entry point: [(ms, units, extras), ...]}."""
        self.enabled = False
        torch.cuda.synchronize()
        if detail:
            return {k: [(a.elapsed_time(b), u, x) for a, b, u, x in v] for k, v in self.events.items()}
        return {k: (len(v), sum(a.elapsed_time(b) for a, b, _, _ in v)) for k, v in self.events.items()}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise IaError(
                f"{_SO} not found: build it with `python -m intrinsicavatar_amd.build` "
                "(there is no CPU fallback for the MI355X hot path)")
        cdll = C.CDLL(_SO)
        cdll.ia_last_error.restype = C.c_char_p
        cdll.ia_scan_tmp_bytes.restype = C.c_int64
        cdll.ia_scan_tmp_bytes.argtypes = [C.c_int64]
        cdll.ia_hashgrid_n_entries.restype = C.c_int64
        cdll.ia_traverse_scratch_bytes.restype = C.c_int64
        cdll.ia_occgrid_tmp_bytes.restype = C.c_int64
        cdll.ia_hashgrid_bwd_scratch_bytes.restype = C.c_int64
        cdll.ia_traverse_fused_scratch_bytes.restype = C.c_int64
        cdll.ia_hashgrid_fwd_scratch_bytes.restype = C.c_int64
        cdll.ia_hashgrid_fwd_levels_jac_offset.restype = C.c_int64
        cdll.ia_eikonal_partials.restype = C.c_int64
        cdll.ia_deform_filter_compact_tmp_bytes.restype = C.c_size_t
        cdll.ia_deform_filter_tiles_tmp_bytes.restype = C.c_size_t
        cdll.ia_deform_filter_tiles_tmp_bytes.argtypes = [C.c_int64]
        cdll.ia_pbr_shade_bwd_scratch_bytes.restype = C.c_size_t
        cdll.ia_morton_order_tmp_bytes.restype = C.c_size_t
        cdll.ia_morton_order_tmp_bytes.argtypes = [C.c_int64]
        cdll.ia_pbr_shade_bwd_scratch_bytes.argtypes = [C.c_int64]
        cdll.ia_deform_filter_compact_tmp_bytes.argtypes = [C.c_int64]
        _lib = _Timed(cdll)
    return _lib


def check(rc: int, what: str = ""):
    _KEEPALIVE.clear()          # the call that consumed the pointers has enqueued its kernels: stream order protects them now
    if rc != 0:
        msg = lib().ia_last_error().decode(errors="replace")
        raise IaError(f"{what} failed (code {rc}): {msg}")


# tensors whose pointers were handed out most recently: a temporary passed as `ptr(x.contiguous())` must stay alive until the
# ctypes call that consumes the pointer has ENQUEUED its kernel -- otherwise the caching allocator may hand its block to the
# next temporary of the same argument list (after the enqueue, stream order makes reuse safe).  64 > arguments per call.
_KEEPALIVE = collections.deque(maxlen=64)        # cleared by check() after every entry-point call


def ptr(t):
    """device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return C.c_void_p(0)
    _KEEPALIVE.append(t)
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
