"""Drop-in for the part of `tinycudann` the reference uses (models/network_utils.py:6,65,191,479,515;
models/utils.py:12,130):

    Encoding(n_input_dims, encoding_config, dtype=torch.float32)   otype in {HashGrid, SphericalHarmonics}
        .params (flat fp32 nn.Parameter)  .n_input_dims  .n_output_dims
        forward / backward (params AND input) / double backward (what the eikonal loss and the normal-conditioned
        radiance need when the reference calls torch.autograd.grad(sdf, x, create_graph=True), rf/geometry.py:165-172)
    free_temporary_memory()

tiny-cuda-nn is not vendored in the reference tree; semantics follow the published Instant-NGP definitions
(oracle/ia_oracle_field.c).  All compute is libia_amd.so.  Third-order terms (d^2 enc / d x^2 is zero almost
everywhere for linear interpolation) are not propagated, as in tiny-cuda-nn."""
import ctypes as C

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L
from . import fields


def free_temporary_memory():
    """tcnn.free_temporary_memory(): nothing to free -- scratch is owned by torch's caching allocator."""
    return None


def _jac_contract(mode, jac, v, out_cols):
    n, K, _ = jac.shape
    v = v.contiguous().float()
    out = torch.empty((n, out_cols), device=jac.device)
    L.check(L.lib().ia_hashgrid_jac_contract(L.i32(mode), L.i64(n), L.i32(K), L.ptr(jac), L.ptr(v), L.i32(v.stride(0)),
                                             L.ptr(out), L.i32(out_cols), L.stream()), "ia_hashgrid_jac_contract")
    return out


class _HashInputGrad(Function):
    """gx = J(x; params)^T gy  -- differentiable w.r.t. params and gy (this is the double-backward hook)."""

    @staticmethod
    def forward(ctx, x, params, gy, cfg):
        _, jac = fields.hashgrid_forward(x, params, cfg, with_jac=True)
        ctx.save_for_backward(x, params, gy, jac)
        ctx.cfg = cfg
        return _jac_contract(0, jac, gy, 3)

    @staticmethod
    def backward(ctx, ggx):
        x, params, gy, jac = ctx.saved_tensors
        ggx = ggx.contiguous().float()
        g_params = None
        if ctx.needs_input_grad[1]:
            g_params = torch.zeros_like(params)
            fields.hashgrid_backward(x, None, g_params, ctx.cfg, g_jac=gy.contiguous().float(), q=ggx)
        g_gy = _jac_contract(1, jac, ggx, jac.shape[1]) if ctx.needs_input_grad[2] else None
        return None, g_params, g_gy, None


class _HashEncode(Function):
    @staticmethod
    def forward(ctx, x, params, cfg):
        x = x.contiguous().float()
        ctx.save_for_backward(x, params)
        ctx.cfg = cfg
        return fields.hashgrid_forward(x, params, cfg)

    @staticmethod
    def backward(ctx, gy):
        x, params = ctx.saved_tensors
        gy = gy.contiguous().float()
        g_params = None
        if ctx.needs_input_grad[1]:
            g_params = torch.zeros_like(params)
            fields.hashgrid_backward(x, gy, g_params, ctx.cfg)
        gx = _HashInputGrad.apply(x, params, gy, ctx.cfg) if ctx.needs_input_grad[0] else None
        return gx, g_params, None


class _SH4(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        ctx.save_for_backward(x)
        return fields.sh4(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        if not (gy.dtype == torch.float32 and gy.dim() == 2 and gy.stride(1) == 1):      # a column block of a wider gradient row is read in place
            gy = gy.contiguous().float()
        gx = torch.empty_like(x)
        L.check(L.lib().ia_sh4_bwd(L.i64(x.shape[0]), L.ptr(x), C.c_void_p(gy.data_ptr()), L.i32(gy.stride(0)), L.ptr(gx), L.stream()), "ia_sh4_bwd")
        return gx


class Encoding(nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: dict, dtype=torch.float32, seed: int = 1337):
        super().__init__()
        if dtype != torch.float32:
            raise NotImplementedError("the reference constructs every tcnn.Encoding with dtype=torch.float32")
        if n_input_dims != 3:
            raise NotImplementedError("3-D inputs only on the render_step path")
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        otype = self.encoding_config.get("otype")
        if otype in ("HashGrid", "Grid"):
            if self.encoding_config.get("interpolation", "Linear") != "Linear":
                raise NotImplementedError("interpolation: Linear only")
            self.cfg = dict(n_levels=int(self.encoding_config.get("n_levels", 16)),
                            n_features_per_level=int(self.encoding_config.get("n_features_per_level", 2)),
                            log2_hashmap_size=int(self.encoding_config.get("log2_hashmap_size", 19)),
                            base_resolution=int(self.encoding_config.get("base_resolution", 16)),
                            per_level_scale=float(self.encoding_config.get("per_level_scale", 2.0)))
            n = fields.hash_n_entries(self.cfg) * self.cfg["n_features_per_level"]
            g = torch.Generator().manual_seed(seed)
            self.params = nn.Parameter((torch.rand(n, generator=g) * 2 - 1) * 1e-4)       # tcnn: U(-1e-4, 1e-4)
            self.n_output_dims = self.cfg["n_levels"] * self.cfg["n_features_per_level"]
            self._kind = "hash"
        elif otype == "SphericalHarmonics":
            if int(self.encoding_config.get("degree", 4)) != 4:
                raise NotImplementedError("SphericalHarmonics degree 4 only (configs/radiance/progressive_hash_grid.yaml:17-19)")
            self.params = nn.Parameter(torch.zeros(0))
            self.n_output_dims = 16
            self._kind = "sh"
        else:
            raise NotImplementedError(f"tcnn.Encoding otype {otype!r} is not on the render_step path")

    def forward(self, x):
        if self._kind == "hash":
            return _HashEncode.apply(x, self.params, self.cfg)
        return _SH4.apply(x)
