"""CPU ORACLE (test infrastructure, NOT the product): numpy restatement of the BRDF / environment-light
definitions behind `lib.torch_pbr` as the reference calls them (models/intrinsic_avatar.py:755-861,
:292-305).  lib/torch_pbr is an empty submodule in the reference tree, so this file DEFINES the
semantics the HIP kernel (csrc/pbr.hip) is held to -- "parity unpinned" against upstream torch_pbr.
Contracts from the call sites: eval -> (diff[P,1], spec[P,3]) incl. the cosine term;
Lo = kd*Lo_diff + ks*Lo_spec with kd = (1-metallic)*albedo, ks = 1; pdf_scale = H*W/(2 pi^2)."""
import numpy as np


def luminance(rgb):
    return 0.2126 * rgb[..., 0] + 0.7152 * rgb[..., 1] + 0.0722 * rgb[..., 2]


def envlight_pmf(base):
    """update_pdf: probability mass per texel proportional to luminance * sin(theta_row)."""
    H, W, _ = base.shape
    sin_t = np.sin((np.arange(H) + 0.5) * np.pi / H)[:, None]
    w = np.maximum(luminance(base.astype(np.float64)), 0) * sin_t
    return (w / w.sum()).astype(np.float32)


def dir_to_uv(d):
    u = np.arctan2(d[..., 0], -d[..., 2]) / (2 * np.pi) + 0.5
    v = np.arccos(np.clip(d[..., 1], -1, 1)) / np.pi
    return u, v


def uv_to_dir(u, v):
    phi = (u - 0.5) * 2 * np.pi
    th = v * np.pi
    return np.stack([np.sin(th) * np.sin(phi), np.cos(th), -np.sin(th) * np.cos(phi)], -1)


def envlight_sample(pmf, k, u1, u2, u3):
    """k directions proportional to pmf: texel by inverse CDF of the flattened pmf (u1), uniform jitter inside the texel."""
    H, W = pmf.shape
    cdf = np.cumsum(pmf.reshape(-1).astype(np.float64))
    idx = np.minimum(np.searchsorted(cdf, u1 * cdf[-1], side="right"), H * W - 1)
    y, x = idx // W, idx % W
    return uv_to_dir((x + u2) / W, (y + u3) / H).astype(np.float32)


def envlight_eval(base, d):
    H, W, _ = base.shape
    u, v = dir_to_uv(d.astype(np.float32))
    fx, fy = u.astype(np.float32) * W - 0.5, v.astype(np.float32) * H - 0.5
    x0, y0 = np.floor(fx), np.floor(fy)
    ax, ay = (fx - x0)[:, None], (fy - y0)[:, None]
    x0, y0 = x0.astype(int), y0.astype(int)
    x1, y1 = (x0 + 1) % W, np.clip(y0 + 1, 0, H - 1)
    x0, y0 = x0 % W, np.clip(y0, 0, H - 1)
    return ((1 - ax) * (1 - ay) * base[y0, x0] + ax * (1 - ay) * base[y0, x1] + (1 - ax) * ay * base[y1, x0]
            + ax * ay * base[y1, x1]).astype(np.float32)


def envlight_pdf(pmf, d):
    H, W = pmf.shape
    u, v = dir_to_uv(d.astype(np.float32))
    x = np.clip((u.astype(np.float32) * W).astype(int), 0, W - 1)
    y = np.clip((v.astype(np.float32) * H).astype(int), 0, H - 1)
    sin_t = np.sin((y + 0.5) * np.pi / H)
    return (pmf[y, x] * (H * W / (2 * np.pi ** 2)) / np.maximum(sin_t, 1e-8)).astype(np.float32)


def brdf_eval(n, wi, wo, alpha, albedo, metallic):
    """MultiLobe (Lambert + isotropic GGX, Smith separable G, Schlick F) incl. cosine. returns diff [P,1], spec [P,3]."""
    NoL = (n * wo).sum(-1)
    NoV = (n * wi).sum(-1)
    diff = np.where(NoL > 0, NoL / np.pi, 0.0)
    h = wi + wo
    hl = np.linalg.norm(h, axis=-1, keepdims=True)
    h = h / np.maximum(hl, 1e-30)
    NoH = (n * h).sum(-1)
    VoH = np.maximum((wi * h).sum(-1), 0)
    a2 = alpha ** 2
    D = a2 / (np.pi * (NoH ** 2 * (a2 - 1) + 1) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        G1l = 2 * NoL / (NoL + np.sqrt(a2 + (1 - a2) * NoL ** 2))
        G1v = 2 * NoV / (NoV + np.sqrt(a2 + (1 - a2) * NoV ** 2))
        common = D * G1l * G1v / (4 * NoV)
    F0 = 0.04 * (1 - metallic[:, None]) + albedo * metallic[:, None]
    F = F0 + (1 - F0) * ((1 - VoH) ** 5)[:, None]
    ok = (NoL > 0) & (NoV > 0) & (hl[:, 0] >= 1e-12)
    spec = np.where(ok[:, None], common[:, None] * F, 0.0)
    return diff[:, None].astype(np.float32), spec.astype(np.float32)


def pbr_light_shade(normal, albedo, roughness, metallic, view_dirs, light_dirs, tr, ind_rgb, base, pmf, w2s_rot):
    """pbr_light_forward (intrinsic_avatar.py:755-861) after the secondary rays have been traced."""
    F_ = normal.shape[0]
    cos_mask = (normal * light_dirs).sum(-1) > 1e-6
    diff, spec = brdf_eval(normal, -view_dirs, light_dirs, roughness, albedo, metallic)
    t = np.clip(tr, 0, 1)
    dw = light_dirs @ w2s_rot
    dw = dw / np.maximum(np.linalg.norm(dw, axis=-1, keepdims=True), 1e-6)
    em = envlight_eval(base, dw)
    pdf = envlight_pdf(pmf, dw)
    live = cos_mask & (t > 0)
    em = np.where(live[:, None], em, 0.0)
    pdf = np.where(live & (pdf > 0), pdf, 1.0)
    Li = em * t[:, None] + (ind_rgb if ind_rgb is not None else 0.0)
    Ld = np.where(cos_mask[:, None], Li * diff / pdf[:, None], 0.0)
    Ls = np.where(cos_mask[:, None], Li * spec / pdf[:, None], 0.0)
    Lo = (1 - metallic[:, None]) * albedo * Ld + Ls
    return Lo.astype(np.float32), Ld.astype(np.float32), Ls.astype(np.float32)


def brdf_pdf(n, wi, wo, alpha):
    """sampling density of the multi-lobe BRDF: 1/2 cosine hemisphere + 1/2 GGX half-vector sampling."""
    NoL = (n * wo).sum(-1)
    p = np.where(NoL > 0, 0.5 * NoL / np.pi, 0.0)
    h = wi + wo
    hl = np.linalg.norm(h, axis=-1, keepdims=True)
    h = h / np.maximum(hl, 1e-30)
    NoH = (n * h).sum(-1)
    VoH = (wi * h).sum(-1)
    a2 = alpha ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        ps = 0.5 * (a2 / (np.pi * (NoH ** 2 * (a2 - 1) + 1) ** 2)) * NoH / (4 * VoH)
    ok = (NoL > 0) & (hl[:, 0] >= 1e-12) & (NoH > 0) & (VoH > 0)
    return (p + np.where(ok, ps, 0.0)).astype(np.float32)


def frame(n):
    sg = np.copysign(1.0, n[:, 2])
    a = -1.0 / (sg + n[:, 2])
    c = n[:, 0] * n[:, 1] * a
    t = np.stack([1.0 + sg * n[:, 0] ** 2 * a, sg * c, -sg * n[:, 0]], -1)
    b = np.stack([c, sg + n[:, 1] ** 2 * a, -n[:, 1]], -1)
    return t, b


def brdf_sample(n, wi, alpha, u):
    t, b = frame(n.astype(np.float64))
    phi = 2 * np.pi * u[:, 2]
    r, z = np.sqrt(u[:, 1]), np.sqrt(np.maximum(1 - u[:, 1], 0))
    wo_d = (r * np.cos(phi))[:, None] * t + (r * np.sin(phi))[:, None] * b + z[:, None] * n
    a2 = alpha.astype(np.float64) ** 2
    ct = np.sqrt(np.maximum((1 - u[:, 1]) / (1 + (a2 - 1) * u[:, 1]), 0))
    st = np.sqrt(np.maximum(1 - ct * ct, 0))
    h = (st * np.cos(phi))[:, None] * t + (st * np.sin(phi))[:, None] * b + ct[:, None] * n
    wo_s = 2 * (wi * h).sum(-1, keepdims=True) * h - wi
    return np.where((u[:, 0] < 0.5)[:, None], wo_d, wo_s).astype(np.float32)


def uniform_sphere_stratified(n_theta, n_phi, u):
    """emitter.sample_uniform_sphere_stratified restricted to the n_theta x n_phi strata the reference indexes
    (models/intrinsic_avatar.py:680-689): equal-area strata in (cos theta, phi), one jittered direction each; pdf = 1/(4 pi)."""
    i = np.repeat(np.arange(n_theta), n_phi).astype(np.float32)
    j = np.tile(np.arange(n_phi), n_theta).astype(np.float32)
    z = np.float32(1.0) - np.float32(2.0) * (i + u[:, 0]) / np.float32(n_theta)
    phi = np.float32(2.0 * np.pi) * (j + u[:, 1]) / np.float32(n_phi)
    r = np.sqrt(np.maximum(np.float32(1.0) - z * z, 0.0))
    dirs = np.stack([r * np.cos(phi), r * np.sin(phi), z], -1).astype(np.float32)
    return dirs, np.full((n_theta * n_phi, 1), 4.0 * np.pi, np.float32)


def pbr_uniform_light_shade(normal, albedo, roughness, metallic, view_dirs, light_dirs, tr, ind_rgb, base, w2s_rot, inv_pdf):
    """pbr_uniform_light_forward (models/intrinsic_avatar.py:654-753) after the secondary rays have been traced:
    weight = inv_pdf (no light pdf), vis = 2 * transmittance."""
    cos_mask = (normal * light_dirs).sum(-1) > 1e-6
    diff, spec = brdf_eval(normal, -view_dirs, light_dirs, roughness, albedo, metallic)
    t = np.where(cos_mask, np.clip(tr, 0, 1), 0.0)
    dw = light_dirs @ w2s_rot
    dw = dw / np.maximum(np.linalg.norm(dw, axis=-1, keepdims=True), 1e-6)
    em = np.where((cos_mask & (t > 0))[:, None], envlight_eval(base, dw), 0.0)
    Li = em * t[:, None] + (np.where(cos_mask[:, None], ind_rgb, 0.0) if ind_rgb is not None else 0.0)
    Ld = np.where(cos_mask[:, None], Li * diff * inv_pdf[:, None], 0.0)
    Ls = np.where(cos_mask[:, None], Li * spec * inv_pdf[:, None], 0.0)
    Lo = (1 - metallic[:, None]) * albedo * Ld + Ls
    vis = np.repeat((2.0 * t)[:, None], 3, 1)
    return Lo.astype(np.float32), Ld.astype(np.float32), Ls.astype(np.float32), vis.astype(np.float32)


def pbr_mats_shade(normal, albedo, roughness, metallic, view_dirs, out_dirs, tr, ind_rgb, base, w2s_rot):
    """pbr_mats_forward (models/intrinsic_avatar.py:863-948) after the secondary rays have been traced: BRDF-sampled
    directions, weight 1 / scatterer.pdf (pdf <= 0 -> 1); no cosine mask, no clamp of the transmittance."""
    wi = -view_dirs
    diff, spec = brdf_eval(normal, wi, out_dirs, roughness, albedo, metallic)
    pdf = brdf_pdf(normal, wi, out_dirs, roughness)
    pdf = np.where(pdf > 0, pdf, 1.0).astype(np.float32)
    dw = out_dirs @ w2s_rot
    dw = dw / np.maximum(np.linalg.norm(dw, axis=-1, keepdims=True), 1e-6)
    Li = envlight_eval(base, dw) * tr[:, None] + (ind_rgb if ind_rgb is not None else 0.0)
    Ld = Li * diff / pdf[:, None]
    Ls = Li * spec / pdf[:, None]
    Lo = (1 - metallic[:, None]) * albedo * Ld + Ls
    return Lo.astype(np.float32), Ld.astype(np.float32), Ls.astype(np.float32)


def pbr_mis_shade(normal, albedo, roughness, metallic, view_dirs, scatter_dirs, light_dirs, tr2, ind_rgb2, base, pmf, w2s_rot):
    """pbr_mis_forward (models/intrinsic_avatar.py:547-652): both strategies' samples ([scatter | light], 2F rays, tr2 /
    ind_rgb2 in that order) weighted by 1 / (pdf_scatter + pdf_light) (the balance heuristic with the Monte-Carlo pdf
    cancelled; <= 1e-6 -> 0), summed over the two strategies."""
    F_ = normal.shape[0]
    rep = lambda a: np.concatenate([a, a], 0)      # noqa: E731
    wo = np.concatenate([scatter_dirs, light_dirs], 0)
    n2, wi2, a2, r2, m2 = rep(normal), rep(-view_dirs), rep(albedo), rep(roughness), rep(metallic)
    pdf_s = brdf_pdf(n2, wi2, wo, r2)
    dw = wo @ w2s_rot
    dw = dw / np.maximum(np.linalg.norm(dw, axis=-1, keepdims=True), 1e-6)
    pdf_l = envlight_pdf(pmf, dw)
    diff, spec = brdf_eval(n2, wi2, wo, r2, a2, m2)
    Li = envlight_eval(base, dw) * tr2[:, None] + (ind_rgb2 if ind_rgb2 is not None else 0.0)
    tot = pdf_s + pdf_l
    w = np.where(tot > 1e-6, 1.0 / np.where(tot > 1e-6, tot, 1.0), 0.0).astype(np.float32)
    Ld = Li * diff * w[:, None]
    Ls = Li * spec * w[:, None]
    Lo = (1 - m2[:, None]) * a2 * Ld + Ls
    s2 = lambda a: a.reshape(2, F_, 3).sum(0).astype(np.float32)      # noqa: E731
    return s2(Lo), s2(Ld), s2(Ls)
