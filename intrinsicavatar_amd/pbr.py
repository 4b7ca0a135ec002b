"""Physically based shading part of render_step on the MI355X kernels (BASELINE configs 3 / 5):

  EnvironmentLightTensor   lib.torch_pbr emitter as used at models/intrinsic_avatar.py:292-305,777-786,819-833
  pbr_light_shade          pbr_light_forward's BRDF / light evaluation (:796-859)            -> ia_pbr_light_shade
  sample_volume_interaction models/pbr/utils.py:70-229 (K1 resampling + attribute gathers)   -> ia_ray_resampling
  light_shuffle            per-ray permutation of the spp light directions (:1356-1378)

lib/torch_pbr is an empty submodule in the reference tree; semantics are those of oracle/pbr_ref.py
(standard Lambert + GGX multi-lobe BRDF, luminance x sin(theta) importance-sampled equirect light).
Random numbers are explicit inputs."""
import math
import os
from typing import Dict, Optional

import ctypes as C

import torch
from torch import Tensor

from . import _lib as L
from . import lib_nerfacc


def _cfg(config, key, default=None):
    """read `key` from an OmegaConf / dict / attribute-style config (the reference builds every module as cls(config))."""
    if config is None:
        return default
    if isinstance(config, dict):
        return config.get(key, default)
    if hasattr(config, "get"):
        try:
            return config.get(key, default)
        except TypeError:
            pass
    return getattr(config, key, default)


class EnvironmentLightTensor(torch.nn.Module):
    """`envlight-tensor` emitter (models/__init__.py:39; configs/light/envlight_tensor.yaml): equirectangular radiance image
    `base` [H,W,3] (a Parameter; the test path replaces it with the HDRI, models/intrinsic_avatar.py:297-301), `.pdf_scale`,
    methods update_pdf / sample / eval / pdf / generate_image / sample_uniform_sphere_stratified.
    Construct from a config (`cls(config)`: envlight_config.{base_res, scale, bias}; random init = scale * U[0,1) + bias, the
    nvdiffrecmc convention -- lib/torch_pbr is absent from the reference tree) or directly from an image tensor."""

    def __init__(self, config=None):
        super().__init__()
        if isinstance(config, Tensor):
            base = config.detach().contiguous().float()
            self.config = None
        else:
            self.config = config
            ec = _cfg(config, "envlight_config", None)
            res = int(_cfg(ec, "base_res", 256))
            scale, bias = float(_cfg(ec, "scale", 0.5)), float(_cfg(ec, "bias", 0.25))
            g = torch.Generator().manual_seed(int(_cfg(ec, "seed", 0)))
            base = torch.rand((res, 2 * res, 3), generator=g) * scale + bias
        self.base = torch.nn.Parameter(base)
        self.pmf = None
        self._cdf = None
        self._pdf_key = None

    def __setattr__(self, name, value):
        if name == "base" and isinstance(value, Tensor) and not isinstance(value, torch.nn.Parameter):
            value = torch.nn.Parameter(value.detach().contiguous().float())
        super().__setattr__(name, value)

    def _pdf_tables(self):
        """(pmf, cdf) of the CURRENT `base`: built on first use and rebuilt whenever `base` was replaced, moved to another
        device or written in place (load_state_dict, .to(), the test path's HDRI swap, an optimiser step) -- the kernels never
        see a NULL or stale table.  update_pdf() forces a rebuild (the training path calls it every step)."""
        b = self.base
        key = (b.data_ptr(), b._version, str(b.device), tuple(b.shape))
        if self.pmf is None or self._pdf_key != key:
            self.update_pdf()
        return self.pmf, self._cdf

    @property
    def pdf_scale(self) -> float:
        return self.base.shape[0] * self.base.shape[1] / (2 * math.pi * math.pi)

    @pdf_scale.setter
    def pdf_scale(self, value):          # assigned by the reference's prepare() (:298-300); derived from the image here
        pass

    def generate_image(self) -> Tensor:
        return self.base

    @torch.no_grad()
    def sample_uniform_sphere_stratified(self, n_rays: int, n_theta: int = 16, n_phi: int = 32, device=None, u: Optional[Tensor] = None):
        """(dirs [n_theta*n_phi, 3], inv_pdf [n_theta*n_phi, 1]): one jittered direction per equal-area stratum -- the set
        the reference indexes with shuffled indices in [0, n_theta*n_phi) (:680-689, :1393-1401).  u [K,2]: explicit jitter."""
        dev = device if device is not None else self.base.device
        if u is None:
            u = torch.rand((n_theta * n_phi, 2), device=dev)
        return uniform_sphere_stratified(n_theta, n_phi, u.to(dev))

    PDF_KERNEL_MAX_PIXELS = 1 << 22          # 4096 tiles of 1024 texels (ia_envlight_pdf_tables); a 1024 x 2048 HDRI has 2^21

    @torch.no_grad()
    def update_pdf(self):
        H, W, _ = self.base.shape
        base = self.base.detach()
        # pmf = luminance x sin(theta), normalised in double; cdf = running sum of the fp32 pmf in double: three tile-parallel launches
        # (the training path rebuilds the tables every step, :777-781)
        if H * W <= self.PDF_KERNEL_MAX_PIXELS:
            self.pmf = torch.empty((H, W), device=base.device)
            self._cdf = torch.empty(H * W, dtype=torch.float64, device=base.device)
            tmp = torch.empty(int(L.lib().ia_envlight_pdf_tables_tmp_bytes(L.i32(H), L.i32(W))), dtype=torch.uint8, device=base.device)
            L.check(L.lib().ia_envlight_pdf_tables(L.i32(H), L.i32(W), L.ptr(base), L.ptr(self.pmf), L.ptr(self._cdf), L.ptr(tmp), L.stream()),
                    "ia_envlight_pdf_tables")
        else:        # larger than the kernel's tile table: the torch expression
            sin_t = torch.sin((torch.arange(H, device=base.device) + 0.5) * math.pi / H)[:, None]
            lum = (0.2126 * base[..., 0] + 0.7152 * base[..., 1] + 0.0722 * base[..., 2]).clamp_min(0).double()
            w = lum * sin_t
            self.pmf = (w / w.sum()).float().contiguous()
            self._cdf = torch.cumsum(self.pmf.reshape(-1).double(), 0)
        self._pdf_key = (self.base.data_ptr(), self.base._version, str(self.base.device), tuple(self.base.shape))

    @torch.no_grad()
    def sample(self, k: int, u: Optional[Tensor] = None, w2s_rot: Optional[Tensor] = None) -> Tensor:
        """k world-space directions proportional to luminance x sin(theta); u [k,3] uniforms (explicit RNG; drawn on the
        device when None).  w2s_rot [3,3]: also apply transform_dirs_w2s (rotate into SMPL space + normalise) in the same
        kernel (ia_envlight_sample)."""
        H, W, _ = self.base.shape
        dev = self.base.device
        if u is None:
            u = torch.rand((k, 3), device=dev)
        u = u.to(dev).float().contiguous()
        out = torch.empty((k, 3), device=dev)
        rot = None if w2s_rot is None else w2s_rot.detach().float().contiguous()
        _, cdf = self._pdf_tables()
        L.check(L.lib().ia_envlight_sample(L.i64(k), L.ptr(u), L.ptr(cdf), L.i32(H), L.i32(W), L.ptr(rot), L.ptr(out),
                                           L.stream()), "ia_envlight_sample")
        return out

    def _eval(self, d: Tensor, want_rgb: bool, want_pdf: bool):
        d = d.contiguous().float()
        n = d.shape[0]
        rgb = torch.empty((n, 3), device=d.device) if want_rgb else None
        pdf = torch.empty((n,), device=d.device) if want_pdf else None
        H, W, _ = self.base.shape
        pmf = self._pdf_tables()[0] if want_pdf else None
        L.check(L.lib().ia_envlight_eval(L.i64(n), L.ptr(d), L.ptr(self.base.detach()), L.ptr(pmf), L.i32(H), L.i32(W), L.ptr(rgb),
                                         L.ptr(pdf), L.stream()), "ia_envlight_eval")
        return rgb, pdf

    def eval(self, d_world: Tensor) -> Tensor:
        return self._eval(d_world, True, False)[0]

    def pdf(self, d_world: Tensor) -> Tensor:
        return self._eval(d_world, False, True)[1][:, None]


class _SGImage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axis, log_lambda, mu, H, W):
        axis, log_lambda, mu = axis.contiguous().float(), log_lambda.contiguous().float(), mu.contiguous().float()
        K = axis.shape[0]
        out = torch.empty((H, W, 3), device=axis.device)
        L.check(L.lib().ia_sg_image(L.i32(K), L.i32(H), L.i32(W), L.ptr(axis), L.ptr(log_lambda), L.ptr(mu), L.ptr(out), L.stream()), "ia_sg_image")
        ctx.save_for_backward(axis, log_lambda, mu)
        ctx.hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, g_img):
        axis, log_lambda, mu = ctx.saved_tensors
        K = axis.shape[0]
        H, W = ctx.hw
        tmp = torch.empty(int(L.lib().ia_sg_image_bwd_tmp_bytes(L.i32(K))), dtype=torch.uint8, device=axis.device)
        g_axis, g_ll, g_mu = torch.empty_like(axis), torch.empty_like(log_lambda), torch.empty_like(mu)
        L.check(L.lib().ia_sg_image_bwd(L.i32(K), L.i32(H), L.i32(W), L.ptr(axis), L.ptr(log_lambda), L.ptr(mu), L.ptr(g_img.contiguous().float()),
                                        L.ptr(tmp), L.ptr(g_axis), L.ptr(g_ll), L.ptr(g_mu), L.stream()), "ia_sg_image_bwd")
        return g_axis, g_ll, g_mu, None, None


class EnvironmentLightSG(torch.nn.Module):
    """`envlight-SG` (configs/light/envlight_SG.yaml: num_SGs 64, base_res 256): the training-time emitter.  lib/torch_pbr
    is an empty submodule, so the parametrisation is the standard spherical-Gaussian mixture (PhySG / nvdiffrecmc):
        L(d) = sum_k softplus(mu_k) * exp(lambda_k * (d . xi_k - 1)),   xi_k = normalize(axis_k), lambda_k = exp(log_lambda_k)
    rendered into an equirectangular [base_res, 2 base_res, 3] image (`generate_image`, differentiable) that the kernels
    evaluate like any EnvironmentLightTensor (`as_tensor_light()`, `env_base=` of pbr_shade_differentiable)."""

    def __init__(self, num_SGs=64, base_res: int = 256, seed: int = 0):
        super().__init__()
        if not isinstance(num_SGs, int):            # cls(config): configs/light/envlight_SG.yaml
            ec = _cfg(num_SGs, "envlight_config", None)
            self.config = num_SGs
            num_SGs, base_res, seed = int(_cfg(ec, "num_SGs", 64)), int(_cfg(ec, "base_res", 256)), int(_cfg(ec, "seed", 0))
        g = torch.Generator().manual_seed(seed)
        i = torch.arange(num_SGs, dtype=torch.float32) + 0.5                    # Fibonacci sphere: even lobe coverage
        phi = math.pi * (1 + 5 ** 0.5) * i
        z = 1 - 2 * i / num_SGs
        r = torch.sqrt((1 - z * z).clamp_min(0))
        self.axis = torch.nn.Parameter(torch.stack([r * torch.cos(phi), z, r * torch.sin(phi)], -1))
        self.log_lambda = torch.nn.Parameter(torch.full((num_SGs,), math.log(20.0)) + 0.2 * torch.randn(num_SGs, generator=g))
        self.mu = torch.nn.Parameter(torch.full((num_SGs, 3), 0.2) + 0.05 * torch.randn((num_SGs, 3), generator=g))
        self.base_res = base_res

    def _dirs(self, device):
        H, W = self.base_res, 2 * self.base_res
        v = (torch.arange(H, device=device) + 0.5) / H
        u = (torch.arange(W, device=device) + 0.5) / W
        th, ph = (v * math.pi)[:, None], ((u - 0.5) * 2 * math.pi)[None, :]
        return torch.stack([torch.sin(th) * torch.sin(ph), torch.cos(th).expand(H, W), -torch.sin(th) * torch.cos(ph)], -1)

    def generate_image(self) -> Tensor:
        """[H,W,3] image of the lobes, differentiable w.r.t. axis / log_lambda / mu: ia_sg_image / _bwd, one launch each way (the torch
        expression -- two [HW,K] GEMMs and ~30 element-wise launches with their backward -- is generate_image_torch)."""
        if not self.axis.is_cuda:
            raise L.IaError("EnvironmentLightSG.generate_image needs its parameters on the GPU (no CPU fallback; generate_image_torch is the "
                            "torch expression the tests compare the kernels with)")
        return _SGImage.apply(self.axis, self.log_lambda, self.mu, self.base_res, 2 * self.base_res)

    def generate_image_torch(self) -> Tensor:
        d = self._dirs(self.axis.device)                                          # [H,W,3], same convention as the kernels
        xi = torch.nn.functional.normalize(self.axis, dim=-1)
        lam = torch.exp(self.log_lambda)
        w = torch.exp(lam * (d.reshape(-1, 3) @ xi.T - 1.0))                       # [HW,K]
        return (w @ torch.nn.functional.softplus(self.mu)).reshape(d.shape)

    def as_tensor_light(self) -> "EnvironmentLightTensor":
        e = EnvironmentLightTensor(self.generate_image().detach())
        e.update_pdf()
        return e

    # -- the emitter surface the model calls (update_pdf / sample / pdf / eval / base / pdf_scale): sampling and pdf go
    # through the equirect image of the current lobes (regenerated by update_pdf, as the training path calls it every step,
    # :777-781); eval is the closed-form lobe sum, differentiable w.r.t. the lobe parameters
    @torch.no_grad()
    def update_pdf(self):
        self.__dict__["_tl"] = self.as_tensor_light()

    def _light(self) -> "EnvironmentLightTensor":
        if "_tl" not in self.__dict__:
            self.update_pdf()
        return self.__dict__["_tl"]

    @property
    def base(self) -> Tensor:
        return self._light().base

    @property
    def pdf_scale(self) -> float:
        return self._light().pdf_scale

    def sample(self, k: int, u: Optional[Tensor] = None, w2s_rot: Optional[Tensor] = None) -> Tensor:
        return self._light().sample(k, u, w2s_rot=w2s_rot)

    def pdf(self, d_world: Tensor) -> Tensor:
        return self._light().pdf(d_world)

    def eval(self, d_world: Tensor) -> Tensor:
        xi = torch.nn.functional.normalize(self.axis, dim=-1)
        w = torch.exp(torch.exp(self.log_lambda) * (d_world @ xi.T - 1.0))
        return w @ torch.nn.functional.softplus(self.mu)

    def sample_uniform_sphere_stratified(self, n_rays: int, n_theta: int = 16, n_phi: int = 32, device=None, u: Optional[Tensor] = None):
        return self._light().sample_uniform_sphere_stratified(n_rays, n_theta, n_phi, device=device, u=u)


# ----------------------------------------------------------------------------- colour helpers of lib.torch_pbr
def rgb_to_srgb(f: Tensor) -> Tensor:
    """linear -> sRGB OETF (models/intrinsic_avatar.py:18,1626-1637; same formula as models/utils.py:98)."""
    f = f.clamp(0.0, 1.0)
    return torch.where(f <= 0.0031308, f * 12.92, torch.pow(f.clamp_min(0.0031308), 1.0 / 2.4) * 1.055 - 0.055)


def luminance(rgb: Tensor) -> Tensor:
    """Rec. 709 luminance, [..., 3] -> [..., 1] (models/pbr/material.py:9)."""
    return 0.2126 * rgb[..., 0:1] + 0.7152 * rgb[..., 1:2] + 0.0722 * rgb[..., 2:3]


def luma(x: Tensor) -> Tensor:
    """channel mean broadcast back to 3 channels (nvdiffrecmc convention; systems/intrinsic_avatar.py:13)."""
    return ((x[..., 0:1] + x[..., 1:2] + x[..., 2:3]) / 3.0).expand_as(x)


def max_value(x: Tensor) -> Tensor:
    """channel maximum broadcast back to 3 channels (nvdiffrecmc convention)."""
    return torch.max(x, dim=-1, keepdim=True)[0].expand_as(x)


MODES = {"light": 0, "uniform_light": 1, "mis": 2, "mats": 3}


def _pmf_of(emitter) -> Tensor:
    """sampling pmf [H,W] of an emitter's current image (never NULL, never stale: EnvironmentLightTensor._pdf_tables)."""
    tl = emitter._light() if isinstance(emitter, EnvironmentLightSG) else emitter
    return tl._pdf_tables()[0]


def pbr_shade(mode: str, normal, albedo, roughness, metallic, view_dirs, out_dirs, transmittance, indirect_rgb,
              emitter: "EnvironmentLightTensor", w2s_rot, inv_pdf=None):
    """one of the four Monte-Carlo estimators (pbr_{light,uniform_light,mis,mats}_forward) for F shading samples and
    their already-traced secondary rays.  returns (Lo, Lo_diff, Lo_spec[, vis])."""
    F_ = normal.shape[0]
    dev = normal.device
    Lo, Ld, Ls = (torch.empty((F_, 3), device=dev) for _ in range(3))
    vis = torch.empty((F_, 3), device=dev) if mode == "uniform_light" else None
    H, W, _ = emitter.base.shape
    c = lambda t: None if t is None else t.contiguous().float()     # noqa: E731
    L.check(L.lib().ia_pbr_shade(
        L.i32(MODES[mode]), L.i64(F_), L.ptr(c(normal)), L.ptr(c(albedo)), L.ptr(c(roughness.reshape(-1))),
        L.ptr(c(metallic.reshape(-1))), L.ptr(c(view_dirs)), L.ptr(c(out_dirs)), L.ptr(c(transmittance.reshape(-1))),
        L.ptr(c(indirect_rgb)), L.ptr(c(inv_pdf.reshape(-1)) if inv_pdf is not None else None), L.ptr(emitter.base),
        L.ptr(_pmf_of(emitter)), L.i32(H), L.i32(W), L.ptr(c(w2s_rot)), L.ptr(Lo), L.ptr(Ld), L.ptr(Ls), L.ptr(vis), L.stream()),
        "ia_pbr_shade")
    return (Lo, Ld, Ls, vis) if mode == "uniform_light" else (Lo, Ld, Ls)


class _PbrShade(torch.autograd.Function):
    """differentiable light / uniform_light estimator: gradients w.r.t. normal, albedo, roughness, metallic and the
    environment texels (ia_pbr_shade_bwd); directions, transmittance, indirect radiance and weights are constants."""

    @staticmethod
    def forward(ctx, mode, normal, albedo, roughness, metallic, env_base, view_dirs, out_dirs, tr, ind, inv_pdf, env_pmf, w2s_rot):
        c = lambda t: None if t is None else t.detach().contiguous().float()     # noqa: E731
        normal, albedo, view_dirs, out_dirs, ind, w2s_rot = (c(t) for t in (normal, albedo, view_dirs, out_dirs, ind, w2s_rot))
        roughness, metallic, tr = (c(t.reshape(-1)) for t in (roughness, metallic, tr))
        inv_pdf = c(inv_pdf.reshape(-1)) if inv_pdf is not None else None
        env_base = c(env_base)
        F_ = normal.shape[0]
        Lo, Ld, Ls = (torch.empty((F_, 3), device=normal.device) for _ in range(3))
        H, W, _ = env_base.shape
        L.check(L.lib().ia_pbr_shade(L.i32(mode), L.i64(F_), L.ptr(normal), L.ptr(albedo), L.ptr(roughness), L.ptr(metallic),
                                     L.ptr(view_dirs), L.ptr(out_dirs), L.ptr(tr), L.ptr(ind), L.ptr(inv_pdf), L.ptr(env_base),
                                     L.ptr(env_pmf), L.i32(H), L.i32(W), L.ptr(w2s_rot), L.ptr(Lo), L.ptr(Ld), L.ptr(Ls),
                                     L.ptr(None), L.stream()), "ia_pbr_shade")
        ctx.mode = mode
        ctx.save_for_backward(normal, albedo, roughness, metallic, env_base, view_dirs, out_dirs, tr, ind, inv_pdf, env_pmf, w2s_rot)
        ctx.set_materialize_grads(False)
        return Lo, Ld, Ls

    @staticmethod
    def backward(ctx, g_Lo, g_Ld, g_Ls):
        normal, albedo, roughness, metallic, env_base, view_dirs, out_dirs, tr, ind, inv_pdf, env_pmf, w2s_rot = ctx.saved_tensors
        F_ = normal.shape[0]
        dev = normal.device
        c = lambda t: None if t is None else t.contiguous().float()     # noqa: E731
        g_n, g_a = torch.empty((F_, 3), device=dev), torch.empty((F_, 3), device=dev)
        g_r, g_m = torch.empty(F_, device=dev), torch.empty(F_, device=dev)
        g_base = L.zeros_like(env_base) if ctx.needs_input_grad[5] else None
        H, W, _ = env_base.shape
        nb = int(L.lib().ia_pbr_shade_bwd_scratch_bytes(L.i64(F_))) if g_base is not None else 0
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None
        L.check(L.lib().ia_pbr_shade_bwd(
            L.i32(ctx.mode), L.i64(F_), L.ptr(normal), L.ptr(albedo), L.ptr(roughness), L.ptr(metallic), L.ptr(view_dirs),
            L.ptr(out_dirs), L.ptr(tr), L.ptr(ind), L.ptr(inv_pdf), L.ptr(env_base), L.ptr(env_pmf), L.i32(H), L.i32(W),
            L.ptr(w2s_rot), L.ptr(c(g_Lo)), L.ptr(c(g_Ld)), L.ptr(c(g_Ls)), L.ptr(g_n), L.ptr(g_a), L.ptr(g_r), L.ptr(g_m),
            L.ptr(g_base), L.ptr(scratch), C.c_size_t(nb), L.stream()), "ia_pbr_shade_bwd")
        return (None, g_n, g_a, g_r[:, None], g_m[:, None], g_base, None, None, None, None, None, None, None)


def pbr_shade_differentiable(mode: str, normal, albedo, roughness, metallic, view_dirs, out_dirs, transmittance, indirect_rgb,
                             emitter: "EnvironmentLightTensor", w2s_rot, inv_pdf=None, env_base: Optional[Tensor] = None):
    """training form of pbr_shade (modes 'light' and 'uniform_light'); roughness / metallic are [F,1];
    env_base: a differentiable [H,W,3] image (e.g. generated from SG lobes) to evaluate instead of emitter.base."""
    assert mode in ("light", "uniform_light")
    base = emitter.base if env_base is None else env_base
    return _PbrShade.apply(MODES[mode], normal, albedo, roughness, metallic, base, view_dirs, out_dirs, transmittance,
                           indirect_rgb, inv_pdf, _pmf_of(emitter), w2s_rot)


def brdf_sample(normal, view_dirs, roughness, u):
    """scatterer.sample: out directions from the multi-lobe BRDF (u [F,3] uniforms)."""
    F_ = normal.shape[0]
    out = torch.empty((F_, 3), device=normal.device)
    L.check(L.lib().ia_brdf_sample(L.i64(F_), L.ptr(normal.contiguous().float()), L.ptr(view_dirs.contiguous().float()),
                                   L.ptr(roughness.reshape(-1).contiguous().float()), L.ptr(u.contiguous().float()), L.ptr(out),
                                   L.stream()), "ia_brdf_sample")
    return out


def brdf_pdf(normal, view_dirs, out_dirs, roughness):
    F_ = normal.shape[0]
    out = torch.empty((F_,), device=normal.device)
    L.check(L.lib().ia_brdf_pdf(L.i64(F_), L.ptr(normal.contiguous().float()), L.ptr(view_dirs.contiguous().float()),
                                L.ptr(out_dirs.contiguous().float()), L.ptr(roughness.reshape(-1).contiguous().float()),
                                L.ptr(out), L.stream()), "ia_brdf_pdf")
    return out[:, None]


def uniform_sphere_stratified(n_theta: int, n_phi: int, u: Tensor):
    """emitter.sample_uniform_sphere_stratified restricted to the 16x32 stratum set that the reference actually indexes
    (shuffled indices are in [0, 512): intrinsic_avatar.py:1393-1401,680-689).  Equal-area strata in (cos theta, phi),
    one jittered direction per stratum (u [n_theta*n_phi, 2]); pdf = 1/(4 pi).  returns (dirs [K,3], inv_pdf [K,1])."""
    dev = u.device
    K = n_theta * n_phi
    if u.shape[0] < K or u.shape[1] < 2:
        raise RuntimeError("uniform_sphere_stratified: u must be [n_theta * n_phi, 2]")
    u2 = u[:K, :2].contiguous().float()
    dirs, inv_pdf = torch.empty((K, 3), device=dev), torch.empty((K, 1), device=dev)
    L.check(L.lib().ia_uniform_sphere_stratified(L.i32(n_theta), L.i32(n_phi), L.ptr(u2), L.ptr(dirs), L.ptr(inv_pdf), L.stream()),
            "ia_uniform_sphere_stratified")
    return dirs, inv_pdf


def pbr_light_shade(normal, albedo, roughness, metallic, view_dirs, light_dirs, transmittance, indirect_rgb,
                    emitter: EnvironmentLightTensor, w2s_rot):
    """fused scatterer.eval + emitter.eval/pdf + Lo assembly for F foreground shading samples."""
    F_ = normal.shape[0]
    dev = normal.device
    Lo, Ld, Ls = (torch.empty((F_, 3), device=dev) for _ in range(3))
    H, W, _ = emitter.base.shape
    c = lambda t: None if t is None else t.contiguous().float()     # noqa: E731
    L.check(L.lib().ia_pbr_light_shade(
        L.i64(F_), L.ptr(c(normal)), L.ptr(c(albedo)), L.ptr(c(roughness.reshape(-1))), L.ptr(c(metallic.reshape(-1))),
        L.ptr(c(view_dirs)), L.ptr(c(light_dirs)), L.ptr(c(transmittance.reshape(-1))), L.ptr(c(indirect_rgb)),
        L.ptr(emitter.base), L.ptr(_pmf_of(emitter)), L.i32(H), L.i32(W), L.ptr(c(w2s_rot)), L.ptr(Lo), L.ptr(Ld), L.ptr(Ls),
        L.stream()), "ia_pbr_light_shade")
    return Lo, Ld, Ls


class VolumeInteraction:
    """K1 (ray_resampling, cdf.cu:10-215) + the layout of its output (csrc/volint.hip): which re-samples are foreground,
    where each ray's / each source interval's foreground re-samples sit in the ray-major foreground list [F].
    Everything sample_volume_interaction (models/pbr/utils.py:70-229) derives with nonzero / gathers / scatters comes
    out of scans and streaming kernels; the only host read-back is (R, F) -- K1's own total and the foreground count, together."""
    K1_CAPACITY = os.environ.get("IA_K1_CAPACITY", "1") == "1"
    K1_CAPACITY_MAX_SLOTS = int(os.environ.get("IA_K1_CAPACITY_MAX_SLOTS", str(1 << 24)))

    @torch.no_grad()
    def __init__(self, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor, n_rays: int, spp: int, weights: Tensor, sdfs: Tensor):
        dev = ray_indices.device
        self.n_rays, self.spp = n_rays, spp
        self.packed_info = lib_nerfacc.pack_info(ray_indices, n_rays)
        # K1's own size (cdf.cu:183) is not read back on its own: its per-resample outputs are written into n_rays x spp slots, the total R
        # stays on the device and comes back TOGETHER with the foreground count F below (IA_K1_CAPACITY=0: two read-backs, the A/B hook)
        totals = torch.empty(2, dtype=torch.int32, device=dev)          # [R, F], both written by scans
        # (ray BATCHES only: for a whole 540 x 540 frame at spp 1024 the 299 M slots -- a quarter of them used -- cost 3 GiB and ~1 ms per
        #  step, measured, against one read-back of 21: profiles/r06_k1_capacity_ab.jsonl)
        capacity = self.K1_CAPACITY and n_rays * spp <= self.K1_CAPACITY_MAX_SLOTS
        if capacity:
            (self.resampled_packed_info, self.ts, self.offsets, self.sampled_idx, self.fg_counts, self.bg_counts,
             self.surface_idx) = lib_nerfacc.ray_resampling_capacity(self.packed_info, t_starts[:, None], t_ends[:, None], weights.detach(),
                                                                     sdfs.detach(), spp, totals[0:1])
        else:
            (self.resampled_packed_info, self.ts, self.offsets, self.sampled_idx, self.fg_counts, self.bg_counts,
             self.surface_idx) = lib_nerfacc.ray_resampling(self.packed_info, t_starts[:, None], t_ends[:, None], weights.detach(),
                                                            sdfs.detach(), spp)
        self.S = int(weights.shape[0])
        lib, st = L.lib(), L.stream()
        self.fg_ray_cnt = torch.empty(n_rays, dtype=torch.int32, device=dev)
        L.check(lib.ia_vi_layout(L.i64(n_rays), L.i32(spp), L.ptr(self.resampled_packed_info), L.ptr(self.bg_counts),
                                 L.ptr(self.fg_ray_cnt), st), "ia_vi_layout")
        self.fg_start = torch.empty(n_rays, dtype=torch.int32, device=dev)
        tmp = L.scan_tmp(max(n_rays, self.S), dev)
        L.check(lib.ia_exclusive_scan_i32(L.ptr(self.fg_ray_cnt), L.ptr(self.fg_start), L.ptr(totals[1:2]), L.i64(n_rays), L.ptr(tmp), st),
                "scan")
        self.fg_off = torch.empty(self.S, dtype=torch.int32, device=dev)          # per source interval (gather backward)
        tot2 = torch.empty(1, dtype=torch.int32, device=dev)
        L.check(lib.ia_exclusive_scan_i32(L.ptr(self.fg_counts), L.ptr(self.fg_off), L.ptr(tot2), L.i64(self.S), L.ptr(tmp), st), "scan")
        if capacity:
            self.R, self.F = totals.tolist()                            # one read-back for both sizes
            self.ts, self.offsets, self.sampled_idx = self.ts[:self.R], self.offsets[:self.R], self.sampled_idx[:self.R]
        else:
            self.R = int(self.ts.shape[0])
            self.F = int(totals[1].item())
        self.fg_src = self.fg_ray = self.positions = self.view_dirs = None

    def gather(self, rays_o, rays_d, weights, normals, albedo, roughness, metallic):
        """-> (w_fg [F], normals [F,3], albedo [F,3], roughness [F,1], metallic [F,1]) differentiable w.r.t. the five
        per-interval inputs; sets .positions / .view_dirs / .fg_src / .fg_ray (constants)."""
        return _VIGather.apply(self, rays_o, rays_d, weights, normals, albedo, roughness, metallic)

    def composite(self, w_fg, Lo, transmittance, background):
        """rgb_phys [n,3] = sum_fg w Lo + transmittance x background (models/intrinsic_avatar.py:1335-1342,1420-1466)."""
        return _VIComposite.apply(self, w_fg, Lo, transmittance, background)

    @torch.no_grad()
    def index_lists(self, weights: Tensor, transmittance: Tensor, want_weights: bool = True):
        """the reference's (fg_indices [F], bg_indices [R-F], resampled_ray_indices [R], resampled_weights [R])."""
        dev = self.ts.device
        fg = torch.empty(self.F, dtype=torch.int64, device=dev)
        bg = torch.empty(self.R - self.F, dtype=torch.int64, device=dev)
        rri = torch.empty(self.R, dtype=torch.int64, device=dev)
        rw = torch.zeros(self.R, device=dev) if want_weights else None
        L.check(L.lib().ia_vi_indices(L.i64(self.n_rays), L.i32(self.spp), L.ptr(self.resampled_packed_info), L.ptr(self.fg_ray_cnt),
                                      L.ptr(self.fg_start), L.ptr(self.bg_counts), L.ptr(self.sampled_idx), L.ptr(self.fg_counts),
                                      L.ptr(weights.detach().float().contiguous()),
                                      L.ptr(transmittance.detach().reshape(-1).float().contiguous()), L.ptr(fg), L.ptr(bg), L.ptr(rri),
                                      L.ptr(rw), L.stream()), "ia_vi_indices")
        return fg, bg, rri, rw

    @torch.no_grad()
    def shuffle(self, shuffle_u: Tensor) -> Tensor:
        """models/intrinsic_avatar.py:1356-1378: per-ray permutation of [0, spp) (stable argsort of the explicit uniforms
        shuffle_u [n_rays, spp]) restricted to the foreground re-samples -> int32 [F]."""
        out = torch.empty(self.F, dtype=torch.int32, device=self.ts.device)
        u = shuffle_u.float().contiguous()
        assert u.shape == (self.n_rays, self.spp)
        L.check(L.lib().ia_light_shuffle(L.i64(self.n_rays), L.i32(self.spp), L.ptr(self.fg_ray_cnt), L.ptr(self.fg_start), L.ptr(u),
                                         L.ptr(out), L.stream()), "ia_light_shuffle")
        return out


class _VIGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vi, rays_o, rays_d, weights, normals, albedo, roughness, metallic):
        dev = weights.device
        F_ = vi.F
        c = lambda t: t.detach().float().contiguous()      # noqa: E731
        vi.fg_src = torch.empty(F_, dtype=torch.int32, device=dev)
        vi.fg_ray = torch.empty(F_, dtype=torch.int32, device=dev)
        vi.positions, vi.view_dirs = torch.empty((F_, 3), device=dev), torch.empty((F_, 3), device=dev)
        o_n, o_a = torch.empty((F_, 3), device=dev), torch.empty((F_, 3), device=dev)
        o_r, o_m, o_w = torch.empty((F_, 1), device=dev), torch.empty((F_, 1), device=dev), torch.empty(F_, device=dev)
        L.check(L.lib().ia_vi_gather(
            L.i64(vi.n_rays), L.ptr(vi.resampled_packed_info), L.ptr(vi.fg_ray_cnt), L.ptr(vi.fg_start), L.ptr(vi.ts),
            L.ptr(vi.sampled_idx), L.ptr(vi.fg_counts), L.ptr(c(weights)), L.ptr(c(rays_o)), L.ptr(c(rays_d)), L.ptr(c(normals)),
            L.ptr(c(albedo)), L.ptr(c(roughness.reshape(-1))), L.ptr(c(metallic.reshape(-1))), L.ptr(vi.fg_src), L.ptr(vi.fg_ray),
            L.ptr(vi.positions), L.ptr(vi.view_dirs), L.ptr(o_n), L.ptr(o_a), L.ptr(o_r), L.ptr(o_m), L.ptr(o_w), L.stream()),
            "ia_vi_gather")
        ctx.vi = vi
        ctx.shapes = (roughness.shape, metallic.shape)
        return o_w, o_n, o_a, o_r, o_m

    @staticmethod
    def backward(ctx, g_w, g_n, g_a, g_r, g_m):
        vi = ctx.vi
        S, dev = vi.S, vi.ts.device
        c = lambda t: None if t is None else t.float().contiguous()      # noqa: E731
        G_n, G_a = torch.empty((S, 3), device=dev), torch.empty((S, 3), device=dev)
        G_r, G_m, G_w = torch.empty(S, device=dev), torch.empty(S, device=dev), torch.empty(S, device=dev)
        L.check(L.lib().ia_vi_gather_bwd(L.i64(S), L.ptr(vi.fg_counts), L.ptr(vi.fg_off), L.ptr(c(g_n)), L.ptr(c(g_a)),
                                         L.ptr(c(g_r.reshape(-1)) if g_r is not None else None),
                                         L.ptr(c(g_m.reshape(-1)) if g_m is not None else None), L.ptr(c(g_w)), L.ptr(G_n), L.ptr(G_a),
                                         L.ptr(G_r), L.ptr(G_m), L.ptr(G_w), L.stream()), "ia_vi_gather_bwd")
        return None, None, None, G_w, G_n, G_a, G_r.reshape(ctx.shapes[0]), G_m.reshape(ctx.shapes[1])


class _VIComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vi, w_fg, Lo, transmittance, background):
        dev = Lo.device
        w_fg, Lo = w_fg.detach().float().contiguous(), Lo.detach().float().contiguous()
        T = transmittance.detach().reshape(-1).float().contiguous()
        bg = background.detach().float().contiguous()
        rgb = torch.empty((vi.n_rays, 3), device=dev)
        L.check(L.lib().ia_vi_composite(L.i64(vi.n_rays), L.ptr(vi.resampled_packed_info), L.ptr(vi.fg_ray_cnt), L.ptr(vi.fg_start),
                                        L.ptr(vi.bg_counts), L.ptr(w_fg), L.ptr(Lo), L.ptr(T), L.ptr(bg), L.ptr(None), L.ptr(rgb),
                                        L.stream()), "ia_vi_composite")
        ctx.vi = vi
        ctx.t_shape = transmittance.shape
        ctx.save_for_backward(w_fg, Lo, bg)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        vi = ctx.vi
        w_fg, Lo, bg = ctx.saved_tensors
        dev = Lo.device
        g_rgb = g_rgb.float().contiguous()
        g_w = torch.empty(vi.F, device=dev) if ctx.needs_input_grad[1] else None
        g_Lo = torch.empty((vi.F, 3), device=dev) if ctx.needs_input_grad[2] else None
        g_T = torch.empty(vi.n_rays, device=dev) if ctx.needs_input_grad[3] else None
        L.check(L.lib().ia_vi_composite_bwd(L.i64(vi.n_rays), L.i64(vi.F), L.ptr(vi.resampled_packed_info), L.ptr(vi.bg_counts),
                                            L.ptr(vi.fg_ray), L.ptr(w_fg), L.ptr(Lo), L.ptr(bg), L.ptr(None), L.ptr(g_rgb), L.ptr(g_w), L.ptr(g_Lo),
                                            L.ptr(g_T), L.stream()), "ia_vi_composite_bwd")
        return None, g_w, g_Lo, (g_T.reshape(ctx.t_shape) if g_T is not None else None), None


@torch.no_grad()
def secondary_rays(normals: Tensor, positions: Tensor, dirs: Tensor, dir_index: Optional[Tensor] = None):
    """the secondary rays of the light estimators (models/intrinsic_avatar.py:788-803): cosine mask n . d > 1e-6, compacted
    ray list.  dirs [F,3], or a direction table + dir_index int32 [F] (shuffled shared light directions).
    returns (rays_o [M,3], rays_d [M,3], src int32 [M] -> index into the F points, out_dirs [F,3])."""
    F_ = normals.shape[0]
    dev = normals.device
    lib, st = L.lib(), L.stream()
    normals, positions, dirs = normals.detach().float().contiguous(), positions.float().contiguous(), dirs.float().contiguous()
    flag = torch.empty(F_, dtype=torch.int32, device=dev)
    L.check(lib.ia_secondary_mask(L.i64(F_), L.ptr(normals), L.ptr(dirs), L.ptr(dir_index), L.ptr(flag), st), "ia_secondary_mask")
    slot = torch.empty(F_, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)          # written by the scan
    tmp = L.scan_tmp(F_, dev)
    L.check(lib.ia_exclusive_scan_i32(L.ptr(flag), L.ptr(slot), L.ptr(total), L.i64(F_), L.ptr(tmp), st), "scan")
    M = int(total.item())
    ro, rd = torch.empty((M, 3), device=dev), torch.empty((M, 3), device=dev)
    src = torch.empty(M, dtype=torch.int32, device=dev)
    dense = torch.empty((F_, 3), device=dev) if dir_index is not None else None
    L.check(lib.ia_secondary_compact(L.i64(F_), L.ptr(flag), L.ptr(slot), L.ptr(positions), L.ptr(dirs), L.ptr(dir_index), L.ptr(ro),
                                     L.ptr(rd), L.ptr(src), L.ptr(dense), st), "ia_secondary_compact")
    _FLAG_SLOT[id(src)] = (src, flag, slot)          # for scatter_secondary: the dense result is written point by point from (flag, slot)
    while len(_FLAG_SLOT) > 2:                        # (an entry keeps [F] int32 x 2 alive: released on use, two unused ones at most)
        _FLAG_SLOT.pop(next(iter(_FLAG_SLOT)))
    return ro, rd, src, (dense if dir_index is not None else dirs)


_FLAG_SLOT = {}          # id(src) -> (src, flag, slot) of the last few secondary_rays calls (several host threads: plain dict operations)


@torch.no_grad()
def scatter_secondary(F_: int, src: Tensor, tr: Tensor, rgb: Tensor):
    """traced (transmittance [M,1], rgb [M,3]) back into dense [F,1] / [F,3] (zeros for masked points), transmittance
    clamped to [0, 1] (:796-803).  With the (flag, slot) of the secondary_rays call that made `src` at hand every dense element is written
    exactly once (ia_secondary_gather_dense: no [F,4] zero fill -- 1.3 GB per headline step); otherwise zero fill + scatter through src."""
    dev = src.device
    ent = _FLAG_SLOT.pop(id(src), None)
    trc, rgbc = tr.reshape(-1).float().contiguous(), rgb.float().contiguous()
    if ent is not None and ent[0] is src and ent[1].shape[0] == F_:
        d_tr, d_rgb = torch.empty((F_, 1), device=dev), torch.empty((F_, 3), device=dev)
        L.check(L.lib().ia_secondary_gather_dense(L.i64(F_), L.ptr(ent[1]), L.ptr(ent[2]), L.ptr(trc), L.ptr(rgbc), L.ptr(d_tr), L.ptr(d_rgb),
                                                  L.stream()), "ia_secondary_gather_dense")
        return d_tr, d_rgb
    buf = L.zeros(F_ * 4, dev)                     # one fill for both
    d_tr, d_rgb = buf[:F_].view(F_, 1), buf[F_:].view(F_, 3)
    L.check(L.lib().ia_secondary_scatter(L.i64(src.shape[0]), L.ptr(src), L.ptr(trc), L.ptr(rgbc), L.ptr(d_tr), L.ptr(d_rgb), L.stream()),
            "ia_secondary_scatter")
    return d_tr, d_rgb


def sample_volume_interaction(rays_o, rays_d, ray_indices, t_starts, t_ends, n_rays: int, spp: int, transmittance_map,
                              extras: Dict[str, Tensor]):
    """models/pbr/utils.py:70-229, same signature and return tuple: (resampled_packed_info, resampled_ray_indices,
    resampled_weights, fg_indices, bg_indices, resampled_extras).  K1 + csrc/volint.hip kernels; differentiable w.r.t.
    extras' weights / normals / albedo / roughness / metallic through the gather kernels.
    DEVIATION (also in INTEGRATION.md): `resampled_weights` is returned DETACHED -- the reference's carries the weight gradient;
    the training path of this package composites with VolumeInteraction.composite (differentiable in the fg weights and the
    transmittance) instead of the dense [R] weights.  A caller that composites with resampled_weights gets no weight gradient."""
    weights, sdfs = extras["weights"], extras["sdf"]
    vi = VolumeInteraction(ray_indices, t_starts, t_ends, n_rays, spp, weights, sdfs)
    fg_idx, bg_idx, rri, rw = vi.index_lists(weights, transmittance_map)
    ex = {}
    if vi.F > 0:
        w_fg, nrm, alb, rough, metal = vi.gather(rays_o, rays_d, weights, extras["normals"], extras["albedo"], extras["roughness"],
                                                 extras["metallic"])
        src = vi.fg_src.long()
        ex = dict(sdf=sdfs[src], alphas=extras["alphas"][src], dists=(t_ends - t_starts)[:, None][src], positions=vi.positions,
                  normals=nrm, albedo=alb, roughness=rough, metallic=metal, t_dirs=vi.view_dirs)
    else:
        # no foreground re-sample: zero-size tensors under every key, as models/pbr/utils.py:208-219 returns them
        dev, z = rays_o.device, (lambda *s_: torch.zeros(s_, device=rays_o.device))     # noqa: E731
        rw = z(0, 1)
        md = extras["metallic"].shape[-1] if extras["metallic"].dim() > 1 else 1
        ex = dict(sdf=z(0), alphas=z(0), dists=z(0, 1), positions=z(0, 3), normals=z(0, 3), albedo=z(0, 3), roughness=z(0, 1),
                  metallic=z(0, md), t_dirs=z(0, 3))
    return vi.resampled_packed_info, rri, rw, fg_idx, bg_idx, ex


@torch.no_grad()
def light_shuffle(n_rays: int, spp: int, resampled_packed_info: Tensor, fg_indices: Tensor, shuffle_u: Tensor) -> Tensor:
    """intrinsic_avatar.py:1356-1378 with the reference's arguments: independent permutation of [0, spp) per ray (stable
    argsort of explicit uniforms instead of the reference's CPU torch.rand), packed over the rays that own re-samples,
    restricted to the re-samples listed in fg_indices.  The permutations come from ia_light_shuffle (LDS bitonic sort)."""
    dev = shuffle_u.device
    rpi = resampled_packed_info.to(torch.int32).contiguous()
    cnt = torch.where(rpi[:, 1] > 0, torch.full_like(rpi[:, 1], spp), torch.zeros_like(rpi[:, 1])).contiguous()
    start = (torch.cumsum(cnt, 0) - cnt).to(torch.int32).contiguous()
    full = torch.empty(int(cnt.sum()), dtype=torch.int32, device=dev)
    L.check(L.lib().ia_light_shuffle(L.i64(n_rays), L.i32(spp), L.ptr(cnt), L.ptr(start), L.ptr(shuffle_u.float().contiguous()),
                                     L.ptr(full), L.stream()), "ia_light_shuffle")
    return full[fg_indices].long()


# ----------------------------------------------------------------------------- scatterer classes of lib.torch_pbr
class _ScattererEval(torch.autograd.Function):
    """(diff [P,1], spec [P,3]) = BRDF x cosine for the lobe set; differentiable w.r.t. normal, albedo, roughness, metallic.
    Backward = the estimator's backward kernel (ia_pbr_shade_bwd, uniform_light form) with unit incident radiance and unit
    weight, so the derivative code exists once."""

    @staticmethod
    def forward(ctx, lobes, n, wi, wo, alpha, albedo, metallic):
        P = n.shape[0]
        dev = n.device
        keep = [t.detach().float().contiguous() for t in (n, wi, wo, alpha.reshape(-1), albedo, metallic.reshape(-1))]
        diff, spec = torch.empty((P, 1), device=dev), torch.empty((P, 3), device=dev)
        L.check(L.lib().ia_scatterer_eval(L.i64(P), L.i32(lobes), *[L.ptr(t) for t in keep], L.ptr(diff), L.ptr(spec), L.stream()),
                "ia_scatterer_eval")
        ctx.lobes = lobes
        ctx.shapes = (alpha.shape, metallic.shape)
        ctx.save_for_backward(*keep)
        return diff, spec

    @staticmethod
    def backward(ctx, g_diff, g_spec):
        n, wi, wo, alpha, albedo, metallic = ctx.saved_tensors
        if ctx.lobes == 4:
            return (None,) * 7
        P, dev = n.shape[0], n.device
        g_Ld = torch.zeros((P, 3), device=dev)
        if ctx.lobes != 2:
            g_Ld[:, 0] = g_diff.reshape(-1)
        g_Ls = g_spec.float().contiguous() if ctx.lobes != 1 else torch.zeros((P, 3), device=dev)
        ones3, zeros, ones = torch.ones((P, 3), device=dev), torch.zeros(P, device=dev), torch.ones(P, device=dev)
        env, pmf, eye = torch.zeros((1, 1, 3), device=dev), torch.ones((1, 1), device=dev), torch.eye(3, device=dev)
        g_n, g_a, g_r, g_m = (torch.empty((P, 3), device=dev), torch.empty((P, 3), device=dev), torch.empty(P, device=dev),
                              torch.empty(P, device=dev))
        g_env = torch.zeros((1, 1, 3), device=dev)
        view = (-wi).contiguous()
        # Li = 0 * em + ind_rgb = 1, weight = inv_pdf = 1  ->  Lo_diff = diff, Lo_spec = spec
        L.check(L.lib().ia_pbr_shade_bwd(L.i32(1), L.i64(P), L.ptr(n), L.ptr(albedo), L.ptr(alpha), L.ptr(metallic), L.ptr(view),
                                         L.ptr(wo), L.ptr(zeros), L.ptr(ones3), L.ptr(ones), L.ptr(env), L.ptr(pmf), L.i32(1), L.i32(1),
                                         L.ptr(eye), L.ptr(None), L.ptr(g_Ld), L.ptr(g_Ls), L.ptr(g_n), L.ptr(g_a), L.ptr(g_r),
                                         L.ptr(g_m), L.ptr(g_env), L.ptr(None), C.c_size_t(0), L.stream()), "ia_pbr_shade_bwd")
        if ctx.lobes == 1:       # the cosine lobe does not depend on the material
            g_a, g_r, g_m = torch.zeros_like(g_a), torch.zeros_like(g_r), torch.zeros_like(g_m)
        return None, g_n, None, None, g_r.reshape(ctx.shapes[0]), g_a, g_m.reshape(ctx.shapes[1])


class _Scatterer(torch.nn.Module):
    """common surface of the lib.torch_pbr scatterers as the reference calls them (keyword arguments, per-point tensors):
        sample(n=, wi=, alpha_x=, alpha_y=, albedo=, metallic=, attenuation=) -> wo [P,3]     (:566-574, :882-890)
        pdf(n=, wi=, wo=, ...)                                                -> [P,1]         (:591-600, :899-908)
        eval(wi=, n=, wo=, ...)                                               -> (diff [P,1], spec [P,3]) incl. cosine (:605-614)
    wi points away from the surface.  Isotropic: alpha_y is accepted and must equal alpha_x (every call site passes the same
    tensor twice); `attenuation` is accepted and ignored ("no attenuation for now", :560-562).  `u` [P,3]: explicit uniforms
    for sample() (drawn on the device when None)."""
    LOBES = 3

    def __init__(self, config=None):
        super().__init__()
        self.config = config

    @staticmethod
    def _alpha(alpha_x, alpha_y):
        if alpha_y is not None and alpha_y is not alpha_x and alpha_y.data_ptr() != alpha_x.data_ptr():
            if not torch.equal(alpha_x, alpha_y):
                raise NotImplementedError("anisotropic roughness (alpha_x != alpha_y) is not used by the reference")
        return alpha_x.reshape(-1)

    @torch.no_grad()
    def sample(self, n, wi, alpha_x, alpha_y=None, albedo=None, metallic=None, attenuation=None, u: Optional[Tensor] = None):
        P, dev = n.shape[0], n.device
        if u is None:
            u = torch.rand((P, 3), device=dev)
        keep = [t.detach().float().contiguous() for t in (n, wi, self._alpha(alpha_x, alpha_y), u)]
        out = torch.empty((P, 3), device=dev)
        L.check(L.lib().ia_scatterer_sample(L.i64(P), L.i32(self.LOBES), *[L.ptr(t) for t in keep], L.ptr(out), L.stream()),
                "ia_scatterer_sample")
        return out

    @torch.no_grad()
    def pdf(self, n, wi, wo, alpha_x, alpha_y=None, albedo=None, metallic=None, attenuation=None):
        P, dev = n.shape[0], n.device
        keep = [t.detach().float().contiguous() for t in (n, wi, wo, self._alpha(alpha_x, alpha_y))]
        out = torch.empty((P, 1), device=dev)
        L.check(L.lib().ia_scatterer_pdf(L.i64(P), L.i32(self.LOBES), *[L.ptr(t) for t in keep], L.ptr(out), L.stream()),
                "ia_scatterer_pdf")
        return out

    def eval(self, wi, n, wo, alpha_x, alpha_y=None, albedo=None, metallic=None, attenuation=None):
        if metallic is not None and metallic.dim() == 2 and metallic.shape[-1] != 1:
            raise NotImplementedError("3-channel specular albedo (volume scattering, phase-* scatterers) is not built")
        if albedo is None:
            albedo = torch.ones_like(n)
        if metallic is None:
            metallic = torch.zeros((n.shape[0], 1), device=n.device)
        return _ScattererEval.apply(self.LOBES, n, wi, wo, self._alpha(alpha_x, alpha_y), albedo, metallic)


class MultiLobe(_Scatterer):
    """`brdf-multi-lobe` (configs/scatterer/brdf-multi-lobe.yaml): Lambert + GGX, lobes sampled 1/2 : 1/2."""
    LOBES = 3


class Lambertian(_Scatterer):
    """`brdf-lambertian`: the cosine lobe alone."""
    LOBES = 1


class GGX(_Scatterer):
    """`brdf-ggx`: the GGX specular lobe alone (Smith G, Schlick F)."""
    LOBES = 2


class Mirror(_Scatterer):
    """`brdf-mirror`: perfect reflection (discrete direction: pdf = 1)."""
    LOBES = 4


class _NotBuilt(torch.nn.Module):
    """a lib.torch_pbr class the render_step path never instantiates with the shipped configs (configs/light/*.yaml,
    configs/scatterer/brdf-multi-lobe.yaml): the NAME resolves so that models/__init__.py:39-51 registers it; constructing
    it raises, like an unsupported option of the reference would."""

    def __init__(self, config=None):
        super().__init__()
        raise NotImplementedError(f"lib.torch_pbr.{type(self).__name__} is not built (IA_ERR_UNSUPPORTED): the MI355X path covers "
                                  "envlight-tensor / envlight-SG and brdf-{multi-lobe, lambertian, ggx, mirror}")


class EnvironmentLightMLP(_NotBuilt):
    pass


class EnvironmentLightNGP(_NotBuilt):
    pass


class DiffuseSGGX(_NotBuilt):
    pass


class SpecularSGGX(_NotBuilt):
    pass


class MultiLobeSGGX(_NotBuilt):
    pass
