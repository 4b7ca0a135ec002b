"""Physically based shading part of render_step on the MI355X kernels (BASELINE configs 3 / 5):

  EnvironmentLightTensor   lib.torch_pbr emitter as used at models/intrinsic_avatar.py:292-305,777-786,819-833
  pbr_light_shade          pbr_light_forward's BRDF / light evaluation (:796-859)            -> ia_pbr_light_shade
  sample_volume_interaction models/pbr/utils.py:70-229 (K1 resampling + attribute gathers)   -> ia_ray_resampling
  light_shuffle            per-ray permutation of the spp light directions (:1356-1378)

lib/torch_pbr is an empty submodule in the reference tree; semantics are those of oracle/pbr_ref.py
(standard Lambert + GGX multi-lobe BRDF, luminance x sin(theta) importance-sampled equirect light).
Random numbers are explicit inputs."""
import math
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib as L
from . import lib_nerfacc


class EnvironmentLightTensor:
    """`emitter` with attributes .base [H,W,3], .pdf_scale and methods update_pdf / sample / eval / pdf."""

    def __init__(self, base: Tensor):
        self.base = base.contiguous().float()
        self.pdf_scale = self.base.shape[0] * self.base.shape[1] / (2 * math.pi * math.pi)
        self.pmf = None
        self._cdf = None

    @torch.no_grad()
    def update_pdf(self):
        H, W, _ = self.base.shape
        sin_t = torch.sin((torch.arange(H, device=self.base.device) + 0.5) * math.pi / H)[:, None]
        lum = (0.2126 * self.base[..., 0] + 0.7152 * self.base[..., 1] + 0.0722 * self.base[..., 2]).clamp_min(0).double()
        w = lum * sin_t
        self.pmf = (w / w.sum()).float().contiguous()
        self._cdf = torch.cumsum(self.pmf.reshape(-1).double(), 0)

    @torch.no_grad()
    def sample(self, k: int, u: Optional[Tensor] = None) -> Tensor:
        """k world-space directions proportional to luminance x sin(theta); u [k,3] uniforms (explicit RNG)."""
        H, W, _ = self.base.shape
        if u is None:
            u = torch.rand((k, 3), device=self.base.device)
        u = u.to(self.base.device).double()
        idx = torch.searchsorted(self._cdf, u[:, 0] * self._cdf[-1], right=True).clamp(max=H * W - 1)
        y, x = idx // W, idx % W
        uu, vv = (x + u[:, 1]) / W, (y + u[:, 2]) / H
        phi, th = (uu - 0.5) * 2 * math.pi, vv * math.pi
        return torch.stack([torch.sin(th) * torch.sin(phi), torch.cos(th), -torch.sin(th) * torch.cos(phi)], -1).float()

    def _eval(self, d: Tensor, want_rgb: bool, want_pdf: bool):
        d = d.contiguous().float()
        n = d.shape[0]
        rgb = torch.empty((n, 3), device=d.device) if want_rgb else None
        pdf = torch.empty((n,), device=d.device) if want_pdf else None
        H, W, _ = self.base.shape
        L.check(L.lib().ia_envlight_eval(L.i64(n), L.ptr(d), L.ptr(self.base), L.ptr(self.pmf), L.i32(H), L.i32(W), L.ptr(rgb),
                                         L.ptr(pdf), L.stream()), "ia_envlight_eval")
        return rgb, pdf

    def eval(self, d_world: Tensor) -> Tensor:
        return self._eval(d_world, True, False)[0]

    def pdf(self, d_world: Tensor) -> Tensor:
        return self._eval(d_world, False, True)[1][:, None]


class EnvironmentLightSG(torch.nn.Module):
    """`envlight-SG` (configs/light/envlight_SG.yaml: num_SGs 64, base_res 256): the training-time emitter.  lib/torch_pbr
    is an empty submodule, so the parametrisation is the standard spherical-Gaussian mixture (PhySG / nvdiffrecmc):
        L(d) = sum_k softplus(mu_k) * exp(lambda_k * (d . xi_k - 1)),   xi_k = normalize(axis_k), lambda_k = exp(log_lambda_k)
    rendered into an equirectangular [base_res, 2 base_res, 3] image (`generate_image`, differentiable) that the kernels
    evaluate like any EnvironmentLightTensor (`as_tensor_light()`, `env_base=` of pbr_shade_differentiable)."""

    def __init__(self, num_SGs: int = 64, base_res: int = 256, seed: int = 0):
        super().__init__()
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
        d = self._dirs(self.axis.device)                                          # [H,W,3], same convention as the kernels
        xi = torch.nn.functional.normalize(self.axis, dim=-1)
        lam = torch.exp(self.log_lambda)
        w = torch.exp(lam * (d.reshape(-1, 3) @ xi.T - 1.0))                       # [HW,K]
        return (w @ torch.nn.functional.softplus(self.mu)).reshape(d.shape)

    def as_tensor_light(self) -> "EnvironmentLightTensor":
        e = EnvironmentLightTensor(self.generate_image().detach())
        e.update_pdf()
        return e


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
        L.ptr(emitter.pmf), L.i32(H), L.i32(W), L.ptr(c(w2s_rot)), L.ptr(Lo), L.ptr(Ld), L.ptr(Ls), L.ptr(vis), L.stream()),
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
        return Lo, Ld, Ls

    @staticmethod
    def backward(ctx, g_Lo, g_Ld, g_Ls):
        normal, albedo, roughness, metallic, env_base, view_dirs, out_dirs, tr, ind, inv_pdf, env_pmf, w2s_rot = ctx.saved_tensors
        F_ = normal.shape[0]
        dev = normal.device
        c = lambda t: None if t is None else t.contiguous().float()     # noqa: E731
        g_n, g_a = torch.empty((F_, 3), device=dev), torch.empty((F_, 3), device=dev)
        g_r, g_m = torch.empty(F_, device=dev), torch.empty(F_, device=dev)
        g_base = torch.zeros_like(env_base) if ctx.needs_input_grad[5] else None
        H, W, _ = env_base.shape
        L.check(L.lib().ia_pbr_shade_bwd(
            L.i32(ctx.mode), L.i64(F_), L.ptr(normal), L.ptr(albedo), L.ptr(roughness), L.ptr(metallic), L.ptr(view_dirs),
            L.ptr(out_dirs), L.ptr(tr), L.ptr(ind), L.ptr(inv_pdf), L.ptr(env_base), L.ptr(env_pmf), L.i32(H), L.i32(W),
            L.ptr(w2s_rot), L.ptr(c(g_Lo)), L.ptr(c(g_Ld)), L.ptr(c(g_Ls)), L.ptr(g_n), L.ptr(g_a), L.ptr(g_r), L.ptr(g_m),
            L.ptr(g_base), L.stream()), "ia_pbr_shade_bwd")
        return (None, g_n, g_a, g_r[:, None], g_m[:, None], g_base, None, None, None, None, None, None, None)


def pbr_shade_differentiable(mode: str, normal, albedo, roughness, metallic, view_dirs, out_dirs, transmittance, indirect_rgb,
                             emitter: "EnvironmentLightTensor", w2s_rot, inv_pdf=None, env_base: Optional[Tensor] = None):
    """training form of pbr_shade (modes 'light' and 'uniform_light'); roughness / metallic are [F,1];
    env_base: a differentiable [H,W,3] image (e.g. generated from SG lobes) to evaluate instead of emitter.base."""
    assert mode in ("light", "uniform_light")
    base = emitter.base if env_base is None else env_base
    return _PbrShade.apply(MODES[mode], normal, albedo, roughness, metallic, base, view_dirs, out_dirs, transmittance,
                           indirect_rgb, inv_pdf, emitter.pmf, w2s_rot)


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
    i = torch.arange(n_theta, device=dev).repeat_interleave(n_phi).float()
    j = torch.arange(n_phi, device=dev).repeat(n_theta).float()
    z = 1.0 - 2.0 * (i + u[:, 0]) / n_theta
    phi = 2.0 * math.pi * (j + u[:, 1]) / n_phi
    r = torch.sqrt((1.0 - z * z).clamp_min(0.0))
    dirs = torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1)
    return dirs, torch.full((n_theta * n_phi, 1), 4.0 * math.pi, device=dev)


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
        L.ptr(emitter.base), L.ptr(emitter.pmf), L.i32(H), L.i32(W), L.ptr(c(w2s_rot)), L.ptr(Lo), L.ptr(Ld), L.ptr(Ls),
        L.stream()), "ia_pbr_light_shade")
    return Lo, Ld, Ls


def sample_volume_interaction(rays_o, rays_d, ray_indices, t_starts, t_ends, n_rays: int, spp: int, transmittance_map,
                              extras: Dict[str, Tensor]):
    """models/pbr/utils.py:70-229: spp stratified samples of the un-normalised weight CDF per ray (+ background bin),
    zero-crossing clamp, per-interval counts, gathers of the per-sample attributes.  The re-sampling itself is not
    differentiated (no_grad in the reference too); the gathers and the re-sampled weights are plain torch ops, so under
    autograd the gradients flow back to weights / normals / albedo / roughness / metallic (training, train_phys.py)."""
    weights, sdfs = extras["weights"], extras["sdf"]
    packed_info = lib_nerfacc.pack_info(ray_indices, n_rays)
    rpi, mid, offs, sampled_idx, fg_cnt, bg_cnt, surface_idx = lib_nerfacc.ray_resampling(
        packed_info, t_starts[:, None], t_ends[:, None], weights, sdfs, spp)
    fg_indices = torch.nonzero(offs[:, 0] < 1e4)[:, 0]
    bg_indices = torch.nonzero(offs[:, 0] >= 1e4)[:, 0]
    rri = lib_nerfacc.unpack_info(rpi, mid.shape[0])
    fg_rri, bg_rri = rri[fg_indices], rri[bg_indices]
    fg_sidx = sampled_idx[fg_indices]
    ex = {}
    if fg_sidx.numel() > 0:
        rw = torch.zeros_like(mid[:, 0])
        rw[fg_indices] = weights[fg_sidx] / fg_cnt[fg_sidx].float()
        rw[bg_indices] = transmittance_map[bg_rri][:, 0] / bg_cnt[bg_rri].float()
        t = mid[fg_indices]
        ex = dict(sdf=sdfs[fg_sidx], alphas=extras["alphas"][fg_sidx], dists=(t_ends - t_starts)[:, None][fg_sidx],
                  positions=rays_o[fg_rri] + rays_d[fg_rri] * t, normals=extras["normals"][fg_sidx],
                  albedo=extras["albedo"][fg_sidx], roughness=extras["roughness"][fg_sidx],
                  metallic=extras["metallic"][fg_sidx], t_dirs=rays_d[fg_rri])
    else:
        rw = torch.zeros((0,), device=rays_o.device)
    return rpi, rri, rw, fg_indices, bg_indices, ex


@torch.no_grad()
def light_shuffle(n_rays: int, spp: int, resampled_packed_info: Tensor, fg_indices: Tensor, shuffle_u: Tensor) -> Tensor:
    """intrinsic_avatar.py:1356-1378: independent permutation of [0, spp) per ray (argsort of uniforms, here an explicit
    device tensor instead of the reference's CPU torch.rand), packed to the resampled points, restricted to fg points."""
    col = torch.argsort(shuffle_u, dim=-1, stable=True)                     # [n_rays, spp]; ties by index (the reference leaves them unspecified)
    has = resampled_packed_info[:, 1] > 0                                   # rays that own spp resampled points
    packed = col[has].reshape(-1)                                           # every such ray owns exactly spp points, in order
    return packed[fg_indices]
