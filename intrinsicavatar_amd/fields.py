"""Neural-field queries of the render_step hot path on the MI355X kernels.

Host-side mirrors of the reference modules, with the reference's parameter names so that its
checkpoints load unchanged:

  VolumeSDF             models/rf/geometry.py:107-235   (ProgressiveBandHashGrid + xyz -> VanillaMLP 35->64->13)
  VolumeRefDirRadiance  models/rf/radiance.py:82-135    (hash grid #2 + feat + SH4(reflect) + normal -> 67->64->64->3)
  VolumeMaterial        models/pbr/material.py:13-51    (LipshitzMLP 48->64->64->5)
  LaplaceDensity        models/rf/density.py:19-34

The kernels take EFFECTIVE weights; weight-norm, Lipschitz normalisation, progressive level masks
and the kernels' column order ([hash 32 | xyz 3 | ...]) are folded in here: one ia_effective_weights
launch per linear layer (differentiable: _EffW), cached per parameter epoch for the no-grad queries.
"""
import ctypes as C
import os
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib as L

HASH = dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
            per_level_scale=1.447269237440378)


def hash_n_entries(cfg=HASH) -> int:
    lib = L.lib()
    return int(lib.ia_hashgrid_n_entries(L.i32(cfg["n_levels"]), L.i32(cfg["log2_hashmap_size"]),
                                         L.i32(cfg["base_resolution"]), L.f32(cfg["per_level_scale"])))


def hashgrid_forward(x01: Tensor, params: Tensor, cfg=HASH, with_jac: bool = False, out: Optional[Tensor] = None):
    """x01 [n,3] in [0,1] -> enc [n,32] (+ jac [n,32,3]); `out` may be a wider [n,stride] buffer (cols 0..31 written)."""
    n = x01.shape[0]
    x01 = x01.contiguous().float()
    LF = cfg["n_levels"] * cfg["n_features_per_level"]
    if out is None:
        out = torch.empty((n, LF), dtype=torch.float32, device=x01.device)
    jac = torch.empty((n, LF, 3), dtype=torch.float32, device=x01.device) if with_jac else None
    # measured (tools/hashfwd_probe.py, 4.4 M points): flat 3.44 / 3.78 ms (without / with Jacobian), XCD-partitioned
    # 2.06 / 3.32 ms (fabric traffic 8x lower, L2 hit 0.91; x-neighbour corners fetched as 16-byte pairs)
    # -- but on the training step's real sample distribution (clustered at the surface) the flat kernel does the Jacobian
    # call in 2.72 ms against 3.29 ms, so the Jacobian call stays flat
    method = os.environ.get("IA_HASH_FWD") or ("xcd" if (n >= HASH_FWD_XCD_MIN and not with_jac) else "flat")
    if method == "xcd":
        nb = int(L.lib().ia_hashgrid_fwd_scratch_bytes(L.i64(n), L.i32(cfg["n_levels"]), L.i32(1 if with_jac else 0)))
        scratch = torch.empty(nb, dtype=torch.uint8, device=x01.device)
        L.check(L.lib().ia_hashgrid_fwd_xcd(L.i64(n), L.ptr(x01), L.ptr(params), L.i32(cfg["n_levels"]),
                                            L.i32(cfg["n_features_per_level"]), L.i32(cfg["log2_hashmap_size"]),
                                            L.i32(cfg["base_resolution"]), L.f32(cfg["per_level_scale"]), L.ptr(out),
                                            L.i32(out.stride(0)), L.ptr(jac), L.ptr(scratch), L.stream()), "ia_hashgrid_fwd_xcd")
        return (out, jac) if with_jac else out
    L.check(L.lib().ia_hashgrid_fwd(L.i64(n), L.ptr(x01), L.ptr(params), L.i32(cfg["n_levels"]),
                                    L.i32(cfg["n_features_per_level"]), L.i32(cfg["log2_hashmap_size"]),
                                    L.i32(cfg["base_resolution"]), L.f32(cfg["per_level_scale"]), L.ptr(out),
                                    L.i32(out.stride(0)), L.ptr(jac), L.stream()), "ia_hashgrid_fwd")
    return (out, jac) if with_jac else out


HASH_FWD_XCD_MIN = 1 << 17          # XCD-partitioned forward from 128 k points (tools/hashfwd_sweep.py; Jacobian-free calls only)
HASH_BWD_BINNED_MIN = 1 << 15        # below this the 4-kernel binned path is launch-bound; plain run-merged atomics
HASH_BWD_CHUNK = 1 << 23             # points per binned call (scratch ~1.3 KB / point, 32-bit record offsets)


def hashgrid_backward(x01: Tensor, g_enc: Optional[Tensor], grad_params: Tensor, cfg=HASH,
                      g_jac: Optional[Tensor] = None, q: Optional[Tensor] = None, method: Optional[str] = None,
                      level_mask: int = 0xFFFFFFFF):
    """accumulate d L / d table into grad_params (see include/ia_amd.h ia_hashgrid_bwd / ia_hashgrid_bwd_binned).
    method: None (by batch size; env IA_HASH_BWD overrides) | 'atomic' | 'binned'."""
    n = x01.shape[0]
    method = method or os.environ.get("IA_HASH_BWD") or ("binned" if n >= HASH_BWD_BINNED_MIN else "atomic")
    cargs = (L.i32(cfg["n_levels"]), L.i32(cfg["n_features_per_level"]), L.i32(cfg["log2_hashmap_size"]),
             L.i32(cfg["base_resolution"]), L.f32(cfg["per_level_scale"]))
    if method == "atomic":
        L.check(L.lib().ia_hashgrid_bwd(L.i64(n), L.ptr(x01), *cargs, L.ptr(g_enc),
                                        L.i32(g_enc.stride(0) if g_enc is not None else 0), L.ptr(g_jac),
                                        L.i32(g_jac.stride(0) if g_jac is not None else 0), L.ptr(q), L.ptr(grad_params),
                                        L.stream()), "ia_hashgrid_bwd")
        return
    for c0 in range(0, n, HASH_BWD_CHUNK):
        m = min(HASH_BWD_CHUNK, n - c0)
        nb = int(L.lib().ia_hashgrid_bwd_scratch_bytes(L.i64(m), L.i32(cfg["n_levels"]), L.i32(cfg["log2_hashmap_size"]),
                                                       L.i32(cfg["base_resolution"]), L.f32(cfg["per_level_scale"])))
        scratch = torch.empty(nb, dtype=torch.uint8, device=x01.device)
        sl = lambda t: None if t is None else t[c0:c0 + m]      # noqa: E731  (row slices keep stride / contiguity)
        ge, gj = sl(g_enc), sl(g_jac)
        L.check(L.lib().ia_hashgrid_bwd_binned(
            L.i64(m), L.ptr(sl(x01)), *cargs, L.ptr(ge), L.i32(ge.stride(0) if ge is not None else 0), L.ptr(gj),
            L.i32(gj.stride(0) if gj is not None else 0), L.ptr(sl(q)), L.ptr(grad_params), C.c_uint32(level_mask & 0xFFFFFFFF),
            L.ptr(scratch), L.i64(nb),
            L.stream()), "ia_hashgrid_bwd_binned")


def sh4(d01: Tensor, out: Optional[Tensor] = None) -> Tensor:
    n = d01.shape[0]
    d01 = d01.contiguous().float()
    if out is None:
        out = torch.empty((n, 16), dtype=torch.float32, device=d01.device)
    L.check(L.lib().ia_sh4_fwd(L.i64(n), L.ptr(d01), L.ptr(out), L.i32(out.stride(0)), L.stream()), "ia_sh4_fwd")
    return out


def mlp_forward(kind: int, segs, W1, b1, W2, b2, Wo, bo, out_dim: int, jac=None, xyz_col=0, inv_scale=None,
                want_grad=False):
    """segs: list of (tensor [n,w_total>=w], width, mul, add). Returns y [n,out_dim] (+ grad [n,3])."""
    n = segs[0][0].shape[0]
    dev = segs[0][0].device
    ns = len(segs)
    keep = []
    ptrs = (C.c_void_p * ns)()
    strides = (C.c_int * ns)()
    widths = (C.c_int * ns)()
    muls = (C.c_float * ns)()
    adds = (C.c_float * ns)()
    for i, (t, w, m, a) in enumerate(segs):
        if t.stride(-1) != 1 or t.dtype != torch.float32 or not t.is_cuda:
            raise L.IaError("MLP input segments must be float32 GPU tensors with unit inner stride")
        keep.append(t)
        ptrs[i] = t.data_ptr()
        strides[i] = t.stride(0)
        widths[i] = w
        muls[i] = m
        adds[i] = a
    y = torch.empty((n, out_dim), dtype=torch.float32, device=dev)
    grad = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_grad else None
    inv = (C.c_float * 3)(*[float(v) for v in inv_scale]) if inv_scale is not None else None
    cont = [t.contiguous().float() if t is not None else None for t in (W1, b1, W2, b2, Wo, bo)]
    L.check(L.lib().ia_mlp_fwd(L.i32(kind), L.i64(n), L.i32(ns), ptrs, strides, widths, muls, adds,
                               L.ptr(cont[0]), L.ptr(cont[1]), L.ptr(cont[2]), L.ptr(cont[3]), L.ptr(cont[4]),
                               L.ptr(cont[5]), L.ptr(y), L.i32(out_dim), L.ptr(jac), L.i32(xyz_col), inv, L.ptr(grad),
                               L.stream()), "ia_mlp_fwd")
    return (y, grad) if want_grad else y


# ----------------------------------------------------------------------------- modules
def _weight_norm(g: Tensor, v: Tensor) -> Tensor:
    return g * v / v.norm(dim=1, keepdim=True)


# The effective weights of a head (weight norm / Lipschitz normalisation, level masks, the kernels' column order) are a function of
# the parameters alone, and a training step asks for them once per field query: five no-grad SDF queries + the differentiable pass
# recomputed ~50 elementwise launches each time, 40 % of the ATen launches of a 4096-ray step (profiles/r06_launch_audit_before.json).
# They are cached per "parameter epoch": PARAM_EPOCH counts the in-place updates made BEHIND torch's back (optim.Adam writes the
# parameters through raw pointers, which does not bump Tensor._version); updates made through torch bump _version, which is part of
# the key as well.  Only no-grad requests are served from the cache (a graph-carrying result must not outlive its backward).
PARAM_EPOCH = [0]


def params_changed():
    """to be called by whoever writes parameters through raw device pointers (optim.Adam.step does)."""
    PARAM_EPOCH[0] += 1


def _cache_key(params, *extra):
    return (PARAM_EPOCH[0], tuple((p.data_ptr(), p._version) for p in params), *extra)


def _cached_nograd(mod, name, key, make, keep=()):
    """`make()` evaluated under no_grad once per key; contiguous detached tensors.  keep: tensors whose (data_ptr, _version) is part
    of the key and which are not owned by the module -- the entry holds them, so their storage cannot be reused under the same key."""
    c = mod.__dict__.get("_wcache")
    if c is None:
        c = mod.__dict__["_wcache"] = {}
    hit = c.get(name)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        val = tuple(t.detach().contiguous() for t in make())
    c[name] = (key, val, tuple(keep))
    return val          # (the other tensors of a key are the module's own parameters / buffers: alive as long as the module)


def normalize_points(x: Tensor, center: Tensor, scale: Tensor) -> Tensor:
    """(x - center) / scale + 0.5 in one launch (ia_normalize_points): the [0,1]^3 coordinates of a hash grid."""
    x = x.contiguous().float()
    out = torch.empty_like(x)
    L.check(L.lib().ia_normalize_points(L.i64(x.shape[0]), L.ptr(x), L.ptr(center), L.ptr(scale), L.ptr(out), L.stream()), "ia_normalize_points")
    return out


class _EffW(torch.autograd.Function):
    """ia_effective_weights / _bwd: (g, v) -> the matrix a fused MLP kernel reads.  mode 0 plain, 1 weight norm (g [M,1]), 2 Lipschitz
    (g = softplus^-1 of the bound, [1]); src int32 [N] / mul float [N]: column order and masks (None: identity / ones)."""

    @staticmethod
    def forward(ctx, mode, g, v, src, mul):
        v = v.contiguous()
        gg = g.contiguous() if g is not None else None
        M, N = v.shape
        out = torch.empty((M, N), device=v.device)
        L.check(L.lib().ia_effective_weights(L.i32(mode), L.i32(M), L.i32(N), L.ptr(gg), L.ptr(v), L.ptr(src), L.ptr(mul), L.ptr(out), L.stream()),
                "ia_effective_weights")
        ctx.mode = mode
        ctx.save_for_backward(gg, v, src, mul)
        return out

    @staticmethod
    def backward(ctx, g_out):
        gg, v, src, mul = ctx.saved_tensors
        M, N = v.shape
        g_v = torch.empty_like(v)
        g_g = torch.empty_like(gg) if gg is not None else None
        L.check(L.lib().ia_effective_weights_bwd(L.i32(ctx.mode), L.i32(M), L.i32(N), L.ptr(gg), L.ptr(v), L.ptr(src), L.ptr(mul),
                                                 L.ptr(g_out.contiguous()), L.ptr(g_v), L.ptr(g_g), L.stream()), "ia_effective_weights_bwd")
        return None, g_g, g_v, None, None


def _col_tables(mod, name, src_cols, mul_parts):
    """(src int32 [N], mul float [N]) of a layer's column order, cached on the module per identity / version of the mask tensors."""
    key = tuple((t.data_ptr(), t._version) if isinstance(t, Tensor) else t for t in mul_parts)
    c = mod.__dict__.setdefault("_coltab", {})
    hit = c.get(name)
    if hit is None or hit[0] != key:
        dev = next(t.device for t in mul_parts if isinstance(t, Tensor))
        src = torch.tensor(src_cols, dtype=torch.int32, device=dev)
        with torch.no_grad():
            mul = torch.cat([t.reshape(-1).float() if isinstance(t, Tensor) else torch.ones(int(t), device=dev) for t in mul_parts]).contiguous()
        # the entry keeps its mask tensors ALIVE: (data_ptr, _version) identifies a tensor only as long as its storage cannot be
        # handed to another one (a freed level mask's block comes back for the next level's mask at the same address)
        hit = c[name] = (key, src, mul, tuple(t for t in mul_parts if isinstance(t, Tensor)))
    return hit[1], hit[2]


class _WNLinear(nn.Module):
    """nn.utils.weight_norm(nn.Linear) parameter layout: weight_g [out,1], weight_v [out,in], bias."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.weight_g = nn.Parameter(torch.ones(dim_out, 1))
        self.weight_v = nn.Parameter(torch.zeros(dim_out, dim_in))
        self.bias = nn.Parameter(torch.zeros(dim_out))

    def effective(self, src=None, mul=None):
        """g * v / |v|_row (ia_effective_weights mode 1; differentiable), optionally in another column order with column masks."""
        return _EffW.apply(1, self.weight_g, self.weight_v, src, mul)


class _Linear(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(dim_out, dim_in))
        self.bias = nn.Parameter(torch.zeros(dim_out))


class HashEncoding(nn.Module):
    """`encoding.encoding.params`: flat fp32 table like tcnn.Encoding (network_utils.py:65)."""

    def __init__(self, cfg=HASH, seed: Optional[int] = None):
        super().__init__()
        self.cfg = cfg
        n = hash_n_entries(cfg) * cfg["n_features_per_level"]
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        self.params = nn.Parameter((torch.rand(n, generator=g) * 2 - 1) * 1e-4)   # tcnn default U(-1e-4, 1e-4)
        self.n_output_dims = cfg["n_levels"] * cfg["n_features_per_level"]


class ProgressiveMask:
    """ProgressiveBandHashGrid.update_step (network_utils.py:79-100), non_smooth mode."""

    def __init__(self, n_levels=16, F_=2, start_level=4, start_step=500, update_steps=125):
        self.n_levels, self.F, self.start_level, self.start_step, self.update_steps = n_levels, F_, start_level, start_step, update_steps
        self.current_level = start_level

    def mask(self, global_step: int, device) -> Tensor:
        lvl = min(self.start_level + max(global_step - self.start_step, 0) // self.update_steps, self.n_levels)
        self.current_level = lvl
        key = (lvl, str(device))
        cache = self.__dict__.setdefault("_masks", {})          # one tensor per (level count, device), kept: not one per call
        m = cache.get(key)
        if m is None:
            m = torch.zeros(self.n_levels * self.F, device=device)
            m[: lvl * self.F] = 1.0
            cache[key] = m
        return m

    def level_bits(self) -> int:
        """bit l set = level l active at the step of the last mask() call (for ia_hashgrid_bwd_binned's level_mask)."""
        return (1 << int(self.current_level)) - 1


class VolumeSDF(nn.Module):
    """models/rf/geometry.py:107-235 on the HIP kernels.  forward(points) with points in canonical space."""

    def __init__(self, seed: Optional[int] = 0, sphere_init_radius=0.5):
        super().__init__()
        self.encoding = nn.Module()
        self.encoding.encoding = nn.Module()
        self.encoding.encoding.encoding = HashEncoding(seed=seed)   # state-dict key: encoding.encoding.encoding.params
        self.network = nn.Module()
        self.network.layers = nn.ModuleList([_WNLinear(35, 64), nn.Identity(), _WNLinear(64, 13)])
        self.prog = ProgressiveMask()
        self.global_step = 25000
        self._init(seed, sphere_init_radius)
        self.register_buffer("center", torch.zeros(3), persistent=False)
        self.register_buffer("scale", torch.ones(3), persistent=False)

    def _init(self, seed, radius):
        # VanillaMLP.make_linear sphere init (network_utils.py:219-231) then weight_norm
        g = torch.Generator().manual_seed(seed if seed is not None else 0)
        l0, l2 = self.network.layers[0], self.network.layers[2]
        w0 = torch.zeros(64, 35)
        w0[:, :3] = torch.randn(64, 3, generator=g) * (math.sqrt(2) / math.sqrt(64))
        w2 = torch.randn(13, 64, generator=g) * 0.0001 + math.sqrt(math.pi) / math.sqrt(64)
        with torch.no_grad():
            l0.weight_v.copy_(w0); l0.weight_g.copy_(w0.norm(dim=1, keepdim=True)); l0.bias.zero_()
            l2.weight_v.copy_(w2); l2.weight_g.copy_(w2.norm(dim=1, keepdim=True)); l2.bias.fill_(-radius)

    def prepare_bbox(self, bbox: Tensor):
        self.center = ((bbox[0] + bbox[1]) / 2).to(self.center)
        self.scale = (bbox[1] - bbox[0]).to(self.scale)
        self._inv_scale_host = None

    def inv_scale_host(self):
        """1 / scale as python floats for the kernels' host-side arguments; read back once per bbox, not once per step."""
        if getattr(self, "_inv_scale_host", None) is None:
            self._inv_scale_host = (1.0 / self.scale).tolist()
        return self._inv_scale_host

    def update_step(self, epoch, global_step):
        self.global_step = global_step

    @property
    def grid_params(self):
        return self.encoding.encoding.encoding.params

    _SRC0 = list(range(3, 35)) + [0, 1, 2]                          # reference column order [xyz | hash] -> kernel order [hash | xyz]

    def _effective_weights(self):
        l0, l2 = self.network.layers[0], self.network.layers[2]
        mask = self.prog.mask(self.global_step, l0.weight_v.device)
        src, mul = _col_tables(self, "l0", self._SRC0, (mask, 3))
        return l0.effective(src, mul), l0.bias, l2.effective(), l2.bias

    def effective_weights(self):
        """kernel column order [hash(32) | xyz(3)], level mask folded into W1.  no_grad callers get the cached tensors of this
        parameter epoch (see PARAM_EPOCH)."""
        if torch.is_grad_enabled():
            return self._effective_weights()
        l0, l2 = self.network.layers[0], self.network.layers[2]
        key = _cache_key((l0.weight_g, l0.weight_v, l0.bias, l2.weight_g, l2.weight_v, l2.bias), self.global_step)
        return _cached_nograd(self, "eff", key, self._effective_weights)

    @torch.no_grad()
    def sdf_only(self, points: Tensor, normalized: bool = False) -> Tensor:
        """SDF value alone (feature[:, 0] of VolumeSDF.forward, rf/geometry.py:152-160) for the no-grad coarse queries:
        XCD-partitioned hash gather whose level-major result feeds the software-pipelined value head directly (no [n,32] feature
        rows, no transpose pass, 4 instead of 52 output bytes per point).  EVERY batch size takes this path, so the value of a
        point does not depend on how many other points share its launch (ray-batch sharding invariance); against forward() the
        output layer is summed in another order (last-bit differences).  normalized: `points` are already the hash grid's
        coordinates (points - center) / scale + 0.5."""
        n = points.shape[0]
        if n == 0 or os.environ.get("IA_SDF_ONLY_FUSED", "1") != "1":
            if normalized:
                points = (points - 0.5) * self.scale + self.center
            return self.forward(points, with_grad=False, with_feature=False).contiguous()
        cfg = HASH
        xp = points.contiguous() if normalized else normalize_points(points, self.center, self.scale)
        nb = int(L.lib().ia_hashgrid_fwd_scratch_bytes(L.i64(n), L.i32(cfg["n_levels"]), L.i32(0)))
        scratch = L.scratch("hash_levels", nb, xp.device)
        L.check(L.lib().ia_hashgrid_fwd_xcd(L.i64(n), L.ptr(xp), L.ptr(self.grid_params), L.i32(cfg["n_levels"]),
                                            L.i32(cfg["n_features_per_level"]), L.i32(cfg["log2_hashmap_size"]),
                                            L.i32(cfg["base_resolution"]), L.f32(cfg["per_level_scale"]), L.ptr(None), L.i32(0),
                                            L.ptr(None), L.ptr(scratch), L.stream()), "ia_hashgrid_fwd_xcd")
        W1k, b1, W2, b2 = self.effective_weights()
        sdf = torch.empty(n, device=xp.device)
        L.check(L.lib().ia_sdf_levels_fwd(L.i64(n), L.ptr(scratch), L.ptr(xp), L.ptr(W1k.contiguous()), L.ptr(b1.contiguous()),
                                          L.ptr(W2.contiguous()), L.ptr(b2.contiguous()), L.ptr(sdf), L.stream()), "ia_sdf_levels_fwd")
        return sdf

    @torch.no_grad()
    def forward(self, points: Tensor, with_grad=True, with_feature=True):
        """returns [sdf, (grad), (feature)] like VolumeSDF.forward (eval / no-grad path)."""
        n = points.shape[0]
        if n == 0:
            out = [points.new_empty(0)]
            if with_grad:
                out.append(points.new_empty(0, 3))
            if with_feature:
                out.append(points.new_empty(0, 13))
            return out[0] if len(out) == 1 else out
        xp = normalize_points(points, self.center, self.scale)
        if with_grad and n >= HASH_FWD_XCD_MIN and os.environ.get("IA_SDF_GRAD_LEVELS", "1") == "1":
            # big batches: one-table-at-a-time gather with Jacobian, level-major results straight into the head (no [n,32] rows, no
            # [n,32,3] Jacobian tensor, coalesced Jacobian reads in the gradient epilogue)
            cfg = HASH
            lib, st = L.lib(), L.stream()
            nb = int(lib.ia_hashgrid_fwd_scratch_bytes(L.i64(n), L.i32(cfg["n_levels"]), L.i32(1)))
            scratch = L.scratch("hash_levels_jac", nb, xp.device)
            L.check(lib.ia_hashgrid_fwd_levels(L.i64(n), L.ptr(xp), L.ptr(self.grid_params), L.i32(cfg["n_levels"]),
                                               L.i32(cfg["n_features_per_level"]), L.i32(cfg["log2_hashmap_size"]),
                                               L.i32(cfg["base_resolution"]), L.f32(cfg["per_level_scale"]), L.i32(1), L.ptr(scratch), st),
                    "ia_hashgrid_fwd_levels")
            joff = int(lib.ia_hashgrid_fwd_levels_jac_offset(L.i64(n), L.i32(cfg["n_levels"])))
            W1k, b1, W2, b2 = self.effective_weights()
            y = torch.empty((n, 13), device=xp.device)
            grad = torch.empty((n, 3), device=xp.device)
            inv = (C.c_float * 3)(*[float(v) for v in self.inv_scale_host()])
            L.check(lib.ia_sdf_levels_fwd_grad(L.i64(n), L.ptr(scratch), C.c_void_p(scratch.data_ptr() + joff), L.ptr(xp),
                                               L.ptr(W1k.contiguous()), L.ptr(b1.contiguous()), L.ptr(W2.contiguous()), L.ptr(b2.contiguous()),
                                               L.ptr(y), L.i32(13), inv, L.ptr(grad), st), "ia_sdf_levels_fwd_grad")
            out = [y[:, 0], grad]
            if with_feature:
                out.append(y)
            return out
        if with_grad:
            enc, jac = hashgrid_forward(xp, self.grid_params, with_jac=True)
        else:
            enc, jac = hashgrid_forward(xp, self.grid_params), None
        W1k, b1, W2, b2 = self.effective_weights()
        res = mlp_forward(0, [(enc, 32, 1.0, 0.0), (xp, 3, 2.0, -1.0)], W1k, b1, None, None, W2, b2, 13, jac=jac,
                          xyz_col=32, inv_scale=self.inv_scale_host() if with_grad else None, want_grad=with_grad)
        y, grad = res if with_grad else (res, None)
        out = [y[:, 0]]
        if with_grad:
            out.append(grad)
        if with_feature:
            out.append(y)
        return out[0] if len(out) == 1 else out


class LaplaceDensity(nn.Module):
    """models/rf/density.py:19-34 (elementwise; fused into the alpha computation of the render path)."""

    def __init__(self, beta_init=0.3, beta_min=1e-4):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(beta_init))
        self.beta_min = beta_min

    def get_beta(self):
        return self.beta.abs() + self.beta_min

    def beta_value(self) -> Tensor:
        """|beta| + beta_min as a detached [1] tensor for the no-grad kernels, cached per parameter epoch (see PARAM_EPOCH)."""
        return _cached_nograd(self, "beta", _cache_key((self.beta,)), lambda: (self.get_beta().reshape(1).float(),))[0]

    def forward(self, sdf):
        beta = self.get_beta()
        return torch.reciprocal(beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


class VolumeRefDirRadiance(nn.Module):
    """models/rf/radiance.py:82-135: [hash#2(32)+xyz(3) | feat(13) | SH4(reflect(-d,n))*mask (16) | n_world (3)] -> 64 -> 64 -> 3."""

    def __init__(self, seed: Optional[int] = 1):
        super().__init__()
        self.xyz_encoding = nn.Module()
        self.xyz_encoding.encoding = nn.Module()
        self.xyz_encoding.encoding.encoding = HashEncoding(seed=seed)
        self.network = nn.Module()
        self.network.layers = nn.ModuleList([_Linear(67, 64), nn.Identity(), _Linear(64, 64), nn.Identity(), _Linear(64, 3)])
        g = torch.Generator().manual_seed(seed if seed is not None else 0)
        for i in (0, 2, 4):        # VanillaMLP.make_linear, non-sphere init (network_utils.py:232-234)
            lin = self.network.layers[i]
            w = torch.empty_like(lin.weight)
            bound = math.sqrt(6.0 / w.shape[1])      # kaiming_uniform_(nonlinearity='relu'): gain sqrt(2), bound = gain*sqrt(3/fan_in)
            with torch.no_grad():
                lin.weight.copy_((torch.rand(w.shape, generator=g) * 2 - 1) * bound)
        self.prog = ProgressiveMask()
        self.global_step = 25000
        self.register_buffer("sh_mask", torch.ones(1, 16))
        # checkpoint-key compatibility: the reference's dir_encoding wraps a tcnn SphericalHarmonics encoding whose (empty)
        # `params` vector is part of the state_dict (tests/golden/golden_state_keys.json)
        self.dir_encoding = nn.Module()
        self.dir_encoding.encoding = nn.Module()
        self.dir_encoding.encoding.params = nn.Parameter(torch.zeros(0), requires_grad=False)
        self.register_buffer("center", torch.zeros(3), persistent=False)
        self.register_buffer("scale", torch.ones(3), persistent=False)
        self.start_step, self.full_band_step = 0, 1

    def prepare_bbox(self, bbox):
        self.center = ((bbox[0] + bbox[1]) / 2).to(self.center)
        self.scale = (bbox[1] - bbox[0]).to(self.scale)

    def update_step(self, epoch, global_step):
        self.global_step = global_step
        t = max(global_step - self.start_step, 0.0)      # radiance.py:140-155
        alpha = 4 * t / (self.full_band_step - self.start_step)
        idx = 0
        for deg in range(4):
            w = (1.0 - math.cos(math.pi * min(max(alpha - deg, 0.0), 1.0))) / 2.0
            self.sh_mask[..., idx:idx + deg * 2 + 1] = w
            idx += deg * 2 + 1

    @property
    def grid_params(self):
        return self.xyz_encoding.encoding.encoding.params

    _SRC0 = list(range(3, 35)) + [0, 1, 2] + list(range(35, 67))    # reference order [xyz(3) hash(32) feat(13) sh(16) normal(3)]

    def _effective_weights(self):
        l = self.network.layers
        W1 = l[0].weight
        m = self.prog.mask(self.global_step, W1.device)
        src, mul = _col_tables(self, "l0", self._SRC0, (m, 3, 13, self.sh_mask, 3))
        W1k = _EffW.apply(0, None, W1, src, mul)
        return W1k, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias

    def effective_weights(self):
        """kernel column order [hash(32) | xyz(3) | feat(13) | sh(16) | normal(3)]; masks folded into W1.  no_grad callers get the
        cached tensors of this parameter epoch (see PARAM_EPOCH)."""
        if torch.is_grad_enabled():
            return self._effective_weights()
        l = self.network.layers
        key = _cache_key((l[0].weight, l[0].bias, l[2].weight, l[2].bias, l[4].weight, l[4].bias, self.sh_mask), self.global_step)
        return _cached_nograd(self, "eff", key, self._effective_weights)

    @torch.no_grad()
    def forward(self, points: Tensor, features: Tensor, refl01: Tensor, normal_world: Tensor, return_embedding=False):
        """returns rgb [n,3] (sigmoid applied). `refl01` = (reflect(-view, n)+1)/2 from ia_shade_prep.
        return_embedding: also return (hash features [n,32], normalised coords [n,3]) = rgb_feature for the material net."""
        n = points.shape[0]
        if n == 0:
            e = points.new_empty(0, 3)
            return (e, points.new_empty(0, 32), e) if return_embedding else e
        xp = normalize_points(points, self.center, self.scale)
        enc = hashgrid_forward(xp, self.grid_params)
        sh = sh4(refl01)
        segs = [(enc, 32, 1.0, 0.0), (xp, 3, 2.0, -1.0), (features.contiguous(), 13, 1.0, 0.0), (sh, 16, 1.0, 0.0),
                (normal_world.contiguous(), 3, 1.0, 0.0)]
        rgb = mlp_forward(1, segs, *self.effective_weights(), 3)
        return (rgb, enc, xp) if return_embedding else rgb


class VolumeMaterial(nn.Module):
    """models/pbr/material.py:13-51 with LipshitzMLP (network_utils.py:360-431): [hash#2(32)+xyz(3) | feat(13)] -> 64 -> 64 -> 5,
    sigmoid, affine to albedo / roughness / metallic.  Parameter names follow LipshitzMLP's ParameterLists."""

    def __init__(self, seed: Optional[int] = 2):
        super().__init__()
        self.network = nn.Module()
        g = torch.Generator().manual_seed(seed if seed is not None else 0)
        dims = [(48, 64), (64, 64), (64, 5)]
        ws, bs, cs = [], [], []
        for di, do in dims:                       # torch.nn.Linear default init
            bound = 1.0 / math.sqrt(di)
            w = (torch.rand((do, di), generator=g) * 2 - 1) * bound
            b = (torch.rand((do,), generator=g) * 2 - 1) * bound
            ws.append(nn.Parameter(w)); bs.append(nn.Parameter(b))
            cs.append(nn.Parameter(torch.ones(1) * w.abs().sum(1).max() * 2))      # network_utils.py:380-385
        # the reference registers every weight twice -- as layers[i].weight and as weights_per_layer[i], one Parameter under
        # two names (network_utils.py:365-377) -- and its checkpoints carry both keys; same aliasing here
        lin = []
        for (di, do), w, b in zip(dims, ws, bs):
            l = nn.Linear(di, do)
            l.weight, l.bias = w, b
            lin.append(l)
        self.network.layers = nn.ModuleList(lin)
        self.network.weights_per_layer = nn.ParameterList(ws)
        self.network.biases_per_layer = nn.ParameterList(bs)
        self.network.lipshitz_bound_per_layer = nn.ParameterList(cs)
        self.albedo_scale, self.albedo_bias = 0.77, 0.03
        self.roughness_scale, self.roughness_bias = 0.9, 0.09
        self.metallic_scale, self.metallic_bias = 1.0, 0.0

    def effective_weights(self, hash_mask: Tensor):
        """Lipschitz normalisation (network_utils.py:396-403) + kernel column order [hash(32) | xyz(3) | feat(13)].  no_grad callers
        get the cached tensors of this parameter epoch (see PARAM_EPOCH)."""
        if torch.is_grad_enabled():
            return self._effective_weights(hash_mask)
        n = self.network
        ps = tuple(n.weights_per_layer) + tuple(n.biases_per_layer) + tuple(n.lipshitz_bound_per_layer)
        key = _cache_key(ps, hash_mask.data_ptr(), hash_mask._version)
        return list(_cached_nograd(self, "eff", key, lambda: self._effective_weights(hash_mask), keep=(hash_mask,)))

    _SRC0 = list(range(3, 35)) + [0, 1, 2] + list(range(35, 48))    # reference input order: [xyz(3) hash(32) | feat(13)]

    def _effective_weights(self, hash_mask: Tensor):
        out = []
        for i in range(3):
            w = self.network.weights_per_layer[i]
            src, mul = _col_tables(self, "l0", self._SRC0, (hash_mask, 3, 13)) if i == 0 else (None, None)
            out += [_EffW.apply(2, self.network.lipshitz_bound_per_layer[i], w, src, mul), self.network.biases_per_layer[i]]
        return out

    def lipshitz_bound_full(self) -> Tensor:
        """LipshitzMLP.lipshitz_bound_full (network_utils.py:405-412): product of the softplus'ed per-layer bounds -- the
        `lipshitz_bound` regulariser of the trainer (configs/config.yaml lambda_lipshitz_bound, from step 12500)."""
        out = 1.0
        for c in self.network.lipshitz_bound_per_layer:
            out = out * torch.nn.functional.softplus(c)
        return out

    def regularizations(self, out=None):
        """VolumeMaterial.regularizations (models/pbr/material.py:53-87) + LipshitzMLP.regularizations (:430-431) for the maps
        a training step produces (train_phys.shade_differentiable_phys): means of the smoothness / orientation maps, the
        Gaussian-histogram entropy of log-albedo over the valid rays, and the Lipschitz bound."""
        ret = {"lipshitz_bound": self.lipshitz_bound_full().mean()}
        if out is None:
            return ret
        for key, name in (("normals_orientation_loss_map", "normal_orientation"), ("albedo_smoothness_loss_map", "albedo_smoothness"),
                          ("roughness_smoothness_loss_map", "roughness_smoothness"), ("metallic_smoothness_loss_map", "metallic_smoothness")):
            if key in out:
                ret[name] = out[key].mean()
        if "comp_albedo_full" in out:
            valid = out["rays_valid_phys_full"][..., 0]
            ret["albedo_entropy"] = albedo_entropy(out["comp_albedo_full"][valid])
        return ret

    @torch.no_grad()
    def forward(self, enc2: Tensor, xp2: Tensor, feat: Tensor, hash_mask: Tensor) -> Tensor:
        """-> [n,5] = albedo(3), roughness(1), metallic(1)"""
        if enc2.shape[0] == 0:
            return enc2.new_empty(0, 5)
        segs = [(enc2, 32, 1.0, 0.0), (xp2, 3, 2.0, -1.0), (feat.contiguous(), 13, 1.0, 0.0)]
        m = mlp_forward(2, segs, *self.effective_weights(hash_mask), 5)
        scale = m.new_tensor([self.albedo_scale] * 3 + [self.roughness_scale, self.metallic_scale])
        bias = m.new_tensor([self.albedo_bias] * 3 + [self.roughness_bias, self.metallic_bias])
        return m * scale + bias


class _GaussianHistogram(torch.autograd.Function):
    """GaussianHistogram(bins, min, max, sigma).forward (models/utils.py:133-149) on ia_gaussian_histogram; differentiable
    w.r.t. the samples and sigma (the reference builds sigma = torch.var(channel) inside the graph)."""

    @staticmethod
    def forward(ctx, x, sigma, bins, vmin, vmax):
        x = x.detach().reshape(-1).float().contiguous()
        sg = sigma.detach().reshape(1).float().contiguous()
        out = torch.zeros(bins, device=x.device)
        L.check(L.lib().ia_gaussian_histogram(L.i64(x.shape[0]), L.ptr(x), L.ptr(sg), L.i32(bins), L.f32(vmin), L.f32(vmax), L.ptr(out),
                                              L.stream()), "ia_gaussian_histogram")
        ctx.cfg = (bins, vmin, vmax)
        ctx.save_for_backward(x, sg)
        return out

    @staticmethod
    def backward(ctx, g):
        x, sg = ctx.saved_tensors
        bins, vmin, vmax = ctx.cfg
        gx, gs = torch.empty_like(x), torch.zeros(1, device=x.device)
        L.check(L.lib().ia_gaussian_histogram_bwd(L.i64(x.shape[0]), L.ptr(x), L.ptr(sg), L.i32(bins), L.f32(vmin), L.f32(vmax),
                                                  L.ptr(g.float().contiguous()), L.ptr(gx), L.ptr(gs), L.stream()),
                "ia_gaussian_histogram_bwd")
        return gx, gs.reshape(()), None, None, None


def gaussian_histogram(x: Tensor, sigma: Tensor, bins: int = 15, vmin: float = 0.0, vmax: float = 1.0) -> Tensor:
    return _GaussianHistogram.apply(x, sigma if isinstance(sigma, Tensor) else torch.tensor(float(sigma), device=x.device), bins, vmin, vmax)


def albedo_entropy(albedo: Tensor) -> Tensor:
    """models/pbr/material.py:58-70: per channel, soft histogram (15 bins on [0,1], sigma = var) of log(albedo + 1e-6),
    normalised (+1e-6), entropy summed over the channels."""
    pred = torch.log(albedo + 1e-6)
    total = 0
    for i in range(pred.shape[-1]):
        ch = pred[..., i].contiguous()
        h = gaussian_histogram(ch, torch.var(ch), 15, 0.0, 1.0)
        # (the reference branches on the host, `if hist.sum() > 1e-6`: here the same select stays on the device -- no read-back)
        s = h.sum()
        ok = s > 1e-6
        h = torch.where(ok, h.div(torch.where(ok, s, torch.ones_like(s))) + 1e-6, torch.ones_like(h))
        total = total + torch.sum(-h * torch.log(h))
    return total
