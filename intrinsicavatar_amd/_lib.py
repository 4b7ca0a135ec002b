"""ctypes loader for libia_amd.so -- the ONLY compute backend of this package.

There is deliberately no CPU / PyTorch fallback: if the HIP library is missing or a
tensor is not on a GPU, the call fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("IA_AMD_LIB") or os.path.join(_HERE, "libia_amd.so")     # IA_AMD_LIB: another build of the same library (A/B runs)
_lib = None


class IaError(RuntimeError):
    pass


# optional outputs that change an entry point's compulsory traffic: ia_fuse_broyden(..., x, J_inv, is_valid, fwd_J, stream)
_EXTRAS = {"ia_fuse_broyden": lambda a: dict(I=int(a[2].value), J_inv=bool(a[16].value), fwd_J=bool(a[18].value)),
           "ia_fuse_broyden_spec": lambda a: dict(I=int(a[1].value), J_inv=bool(a[15].value), fwd_J=bool(a[17].value)),
           "ia_fuse_broyden_spec_rows": lambda a: dict(I=int(a[1].value), J_inv=bool(a[15].value), fwd_J=bool(a[16].value), rows=True)}


class _Timed:
    """optional per-entry-point HIP-event timing (bench.py): events are recorded on the stream the
    kernels are launched on (torch's current stream), nothing synchronises until report()."""

    def __init__(self, cdll):
        self._cdll = cdll
        self.enabled = False
        self.events = {}

    def __getattr__(self, name):
        # (called once per entry point: the resolved callable is stored on the instance, later look-ups never get here -- the 4096-ray
        #  training step is bound by the host's launch rate, profiles/r06_launch_audit_before.json)
        fn = self._resolve(name)
        self.__dict__[name] = fn
        return fn

    def _resolve(self, name):
        fn = getattr(self._cdll, name)
        if not name.startswith("ia_") or name in ("ia_last_error", "ia_scan_tmp_bytes", "ia_version", "ia_hashgrid_n_entries", "ia_traverse_scratch_bytes", "ia_occgrid_tmp_bytes", "ia_hashgrid_bwd_scratch_bytes", "ia_traverse_fused_scratch_bytes", "ia_hashgrid_fwd_scratch_bytes", "ia_eikonal_partials", "ia_spec_rows_slots", "ia_spec_rows_overflow_bytes", "ia_spec_rows_overflow_capacity", "ia_resample_tmp_bytes", "ia_sg_image_bwd_tmp_bytes", "ia_envlight_pdf_tables_tmp_bytes", "ia_phys_loss_tmp_bytes", "ia_hashgrid_fwd_levels_jac_offset", "ia_morton_order_tmp_bytes", "ia_deform_filter_tiles_tmp_bytes", "ia_deform_filter_compact_tmp_bytes"):
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


_HEADER = os.path.join(os.path.dirname(_HERE), "include", "ia_amd.h")
_CTYPE = {"int": C.c_int, "int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "uint8_t": C.c_uint8,
          "size_t": C.c_size_t, "float": C.c_float, "double": C.c_double, "ia_stream_t": C.c_void_p, "void": None}


def header_prototypes(path: str = _HEADER):
    """{name: (restype, [argtypes])} of every `ia_*` prototype in include/ia_amd.h -- the header is the single source of the
    ABI: the loader declares argtypes / restype for EVERY entry point from it, so a call with a missing, extra or mistyped
    argument raises in Python instead of reading garbage off the stack.  Pointers (device or host) are void*."""
    import re
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    txt = re.sub(r"^\s*#[^\n]*", " ", txt, flags=re.M)                # preprocessor lines
    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(ia_\w+)\s*\(([^()]*)\)\s*;", txt):
        def ctype(decl, is_ret=False):
            decl = decl.replace("const", " ").strip()
            if "*" in decl:
                return C.c_char_p if (is_ret and "char" in decl) else C.c_void_p
            base = decl.split()[0] if is_ret else decl.split()[0]
            if base not in _CTYPE:
                raise IaError(f"include/ia_amd.h: unknown type in prototype of {name}: {decl!r}")
            return _CTYPE[base]
        ret = ret.strip()
        if ret.startswith("typedef") or not ret:
            continue
        arglist = [a_.strip() for a_ in args.split(",") if a_.strip()]
        if arglist == ["void"]:
            arglist = []
        protos[name] = (ctype(ret, True), [ctype(a_) for a_ in arglist])
    return protos


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise IaError(
                f"{_SO} not found: build it with `python -m intrinsicavatar_amd.build` "
                "(there is no CPU fallback for the MI355X hot path)")
        if not os.path.exists(_HEADER):
            raise IaError(f"{_HEADER} not found: the C-ABI header declares the argument types of libia_amd.so")
        cdll = C.CDLL(_SO)
        for name, (restype, argtypes) in header_prototypes().items():
            fn = getattr(cdll, name)              # AttributeError: the header declares a symbol the library does not export
            fn.restype, fn.argtypes = restype, argtypes
        _lib = _Timed(cdll)
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().ia_last_error().decode(errors="replace")
        raise IaError(f"{what} failed (code {rc}): {msg}")


class _TensorPtr(C.c_void_p):
    """device pointer that OWNS a reference to its tensor.  A temporary passed as `ptr(x.contiguous())` must stay alive
    until the ctypes call that consumes the pointer has ENQUEUED its kernel -- otherwise the caching allocator may hand its
    block to the next temporary of the same argument list (after the enqueue, stream order makes reuse safe).  The pointer
    objects live in the argument tuple of the call, so the tensors live exactly as long as the call: no global state, no
    limit on the number of arguments, nested entry-point calls inside an argument expression cannot release anything."""


def ptr(t):
    """device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise IaError("intrinsicavatar_amd operators need GPU tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise IaError("tensor must be contiguous")
    p = _TensorPtr(t.data_ptr())
    p._keep = t
    return p


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SCRATCH = {}


def scratch(name: str, nbytes: int, device):
    """grow-only scratch buffer `name` of at least `nbytes` bytes on `device` (uint8 view).  For the big per-call work areas that a
    launch consumes before the next call on the same stream can touch them (level-major hash features: 128 B / point; the sort's
    key / index columns): when the batch size changes from call to call -- another pose every frame -- the caching allocator
    cannot reuse its blocks and every call pays a multi-GB hipMalloc / hipFree (2.7 against 0.9 s per relit 540 x 540 frame)."""
    nbytes = int(nbytes)
    key = (name, str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _SCRATCH.get(key)
    # (no automatic trimming: the batches of ONE step differ by 10 x in size, and giving a work area back after a small batch makes the
    #  next large one pay a multi-GB hipMalloc -- measured: +37 ms per headline step; scratch_clear() is the explicit release)
    if buf is None or buf.numel() < nbytes:
        _SCRATCH[key] = None
        del buf
        buf = torch.empty(nbytes + nbytes // 4 + 4096, dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf[:nbytes]


def scratch_clear():
    """release every grow-only scratch buffer (they are views into the binding's own cache: a view must not outlive the call it was
    made for, and nothing else holds them).  Long-running hosts call this after an unusually large batch."""
    _SCRATCH.clear()


def scan_tmp(n: int, device, extra_bytes: int = 0):
    nbytes = int(lib().ia_scan_tmp_bytes(C.c_int64(max(int(n), 1)))) + int(extra_bytes) + 64
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


_ZEROS_FILL = os.environ.get("IA_ZEROS", "fill") == "fill"


def zeros(shape, device, dtype=torch.float32):
    """zero-filled tensor through a fill KERNEL (torch.full) instead of torch.zeros' hipMemsetAsync: on the launch-rate-bound 4096-ray step
    the device idles ~30 us in front of every memset against ~5 us in front of a kernel (profiles/r06_launch_audit_after_stepops.json:
    455 us of idle in front of 15 memsets).  IA_ZEROS=memset restores torch.zeros (A/B)."""
    if _ZEROS_FILL:
        return torch.full(shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,), 0, dtype=dtype, device=device)
    return torch.zeros(shape, dtype=dtype, device=device)


def zeros_like(t):
    return zeros(tuple(t.shape), t.device, t.dtype)


i64 = C.c_int64
i32 = C.c_int
f32 = C.c_float
