"""Plain-PyTorch (CPU, float64) reference of the differentiable shading pass, used ONLY to check the HIP
backward kernels: hash grid, SH4, MLPs, analytic normal via autograd.grad(create_graph=True) exactly as
models/rf/geometry.py:165-172 does, reflect / normalise, Laplace alpha, transmittance weights, accumulation."""
import math

import torch


def hash_cfg(n_levels=16, log2_T=19, base=16, pls=1.447269237440378):
    import numpy as np
    offs, ress, scs, off = [], [], [], 0
    l2 = np.log2(np.float32(pls))
    for l in range(n_levels):
        sc = np.float32(np.exp2(np.float32(l) * l2) * np.float32(base) - np.float32(1.0))
        res = int(np.ceil(sc)) + 1
        p = min((res ** 3 + 7) // 8 * 8, 1 << log2_T)
        offs.append(off); ress.append(res); scs.append(float(sc)); off += p
    offs.append(off)
    return offs, ress, scs


def hashgrid(x01, table, cfg=None):
    """x01 [n,3] float64 (requires_grad ok), table [entries,2] -> [n,32]."""
    offs, ress, scs = cfg or hash_cfg()
    outs = []
    for l in range(len(ress)):
        sc, res, hs = scs[l], ress[l], offs[l + 1] - offs[l]
        pos = x01 * sc + 0.5
        pg = torch.floor(pos).detach()
        w = pos - pg
        pg = pg.long()
        acc = 0
        for c in range(8):
            o = torch.tensor([(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1])
            p = pg + o
            wc = torch.where(o.bool(), w, 1 - w).prod(-1, keepdim=True)
            stride, dense_ok = 1, True
            idx = p[:, 0] * 1
            stride = res
            if stride <= hs:
                idx = idx + p[:, 1] * stride
                stride *= res
                if stride <= hs:
                    idx = idx + p[:, 2] * stride
                    stride *= res
            if hs < stride:
                M = 0xFFFFFFFF
                idx = ((p[:, 0] * 1) & M) ^ ((p[:, 1] * 2654435761) & M) ^ ((p[:, 2] * 805459861) & M)
            idx = (idx & 0xFFFFFFFF) % hs
            acc = acc + wc * table[offs[l] + idx]
        outs.append(acc)
    return torch.cat(outs, -1)


def sh4(d01):
    x, y, z = (d01 * 2 - 1).unbind(-1)
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z,
        -0.48860251190291987 * x, 1.0925484305920792 * xy, -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


class _HashInputGradRef(torch.autograd.Function):
    """J(x)^T g as a function of (g, table) only: its own backward has no x term -- the second derivative of the encoding
    with respect to its input is dropped, as in the kernels (pose gradients)."""

    @staticmethod
    def forward(ctx, g, x, table):
        ctx.save_for_backward(g, x, table)
        with torch.enable_grad():
            x_ = x.detach().requires_grad_(True)
            return torch.autograd.grad(hashgrid(x_, table.detach()), x_, g.detach())[0]

    @staticmethod
    def backward(ctx, v):
        g, x, table = ctx.saved_tensors
        with torch.enable_grad():
            x_ = x.detach().requires_grad_(True)
            t_ = table.detach().requires_grad_(True)
            u = g.detach().requires_grad_(True)
            s = torch.autograd.grad(hashgrid(x_, t_), x_, u, create_graph=True)[0]
            Jv, gt = torch.autograd.grad((s * v.detach()).sum(), (u, t_))
        return Jv, None, gt


class _HashEncRef(torch.autograd.Function):
    """hash encoding whose input gradient is differentiable in (upstream gradient, table) but not in x."""

    @staticmethod
    def forward(ctx, x, table):
        ctx.save_for_backward(x, table)
        return hashgrid(x.detach(), table.detach())

    @staticmethod
    def backward(ctx, g):
        x, table = ctx.saved_tensors
        with torch.enable_grad():
            t_ = table.detach().requires_grad_(True)
            gt = torch.autograd.grad(hashgrid(x.detach(), t_), t_, g.detach())[0]
        return _HashInputGradRef.apply(g, x, table), gt


def shade_reference(P, fixed, target_rgb, target_mask, lambda_eik=0.1, lambda_mask=0.1, lambda_curv=0.0):
    """P: dict of float64 leaf tensors (requires_grad) in the REFERENCE layout; fixed: sample set found by the GPU."""
    x = fixed["pts_cano"].clone().requires_grad_(True)
    valid = fixed["valid"]
    pose = "tfs" in P
    c2w = fixed["c2w"]
    x_in = x
    if pose:      # ForwardDeformer.forward version 1 (deformer_torch.py:57-76) on the winners + blended-rotation push-forward
        T = (fixed["lbs_w"] @ P["tfs"].reshape(-1, 16)).reshape(-1, 4, 4)
        Rb = T[:, :3, :3]
        xd = (Rb * fixed["pts_cano"][:, None, :]).sum(-1) + T[:, :3, 3]
        corr = -(fixed["J_inv"] * (xd - xd.detach())[:, None, :]).sum(-1) * valid[:, None].double()
        x_in = x + corr
        c2w = c2w + (Rb - Rb.detach()) * valid[:, None, None].double()
    # --- VolumeSDF.forward (geometry.py:152-172)
    xp = (x_in - P["geo_center"]) / P["geo_scale"] + 0.5
    if pose:
        enc = _HashEncRef.apply(xp, P["geo_table"].reshape(-1, 2)) * P["geo_mask"]
    else:
        enc = hashgrid(xp, P["geo_table"].reshape(-1, 2)) * P["geo_mask"]
    h = torch.cat([xp * 2 - 1, enc], -1)
    W1 = P["geo_g0"] * P["geo_v0"] / P["geo_v0"].norm(dim=1, keepdim=True)
    W2 = P["geo_g2"] * P["geo_v2"] / P["geo_v2"].norm(dim=1, keepdim=True)
    a = torch.nn.functional.softplus(h @ W1.T + P["geo_b0"], beta=100)
    out = a @ W2.T + P["geo_b2"]
    grad_c = torch.autograd.grad(out[:, 0], x, torch.ones_like(out[:, 0]), create_graph=True)[0]
    laplace = None
    if lambda_curv > 0.0:        # geometry.py:173-203 with the probe point treated as a constant
        nz = torch.nn.functional.normalize
        tang = torch.cross(nz(grad_c.detach(), dim=-1, eps=1e-6), nz(fixed["curv_u"], dim=-1, eps=1e-6), dim=-1)
        x_d = (x.detach() + 1e-4 * tang).requires_grad_(True)
        xp_d = (x_d - P["geo_center"]) / P["geo_scale"] + 0.5
        h_d = torch.cat([xp_d * 2 - 1, hashgrid(xp_d, P["geo_table"].reshape(-1, 2)) * P["geo_mask"]], -1)
        sdf_d = (torch.nn.functional.softplus(h_d @ W1.T + P["geo_b0"], beta=100) @ W2.T + P["geo_b2"])[:, 0]
        grad_d = torch.autograd.grad(sdf_d, x_d, torch.ones_like(sdf_d), create_graph=True)[0]
        dot = (nz(grad_c, dim=-1, eps=1e-6) * nz(grad_d, dim=-1, eps=1e-6)).sum(-1)
        laplace = torch.acos(dot.clamp(-1 + 1e-6, 1 - 1e-6)) / math.pi * valid.double()
    vf = valid[:, None].double()
    feat = out * vf
    sdf = torch.where(valid, out[:, 0], torch.full_like(out[:, 0], 1e5))
    sdf_grad = torch.where(valid[:, None], torch.einsum("bij,bj->bi", c2w, grad_c), torch.tensor([0., 0., 1.], dtype=x.dtype))
    # --- shade prep
    R = fixed["w2s_rot"]
    nrm = lambda v: v / v.norm(dim=-1, keepdim=True).clamp_min(1e-6)     # noqa: E731
    nw = nrm(sdf_grad @ R)
    vw = nrm(fixed["rays_d"][fixed["ray_indices"]] @ R)
    xx = -vw
    refl = 2 * (xx * nw).sum(-1, keepdim=True) * nw - xx
    # --- alpha (density.py:25-30, intrinsic_avatar.py:390-394)
    beta = P["beta"].abs() + 1e-4
    dens = (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))
    alphas = 1 - torch.exp(-dens * (fixed["t_ends"] - fixed["t_starts"]))
    # --- radiance (radiance.py:111-135)
    xp2 = ((x_in if pose else x.detach()) - P["rad_center"]) / P["rad_scale"] + 0.5
    enc2 = hashgrid(xp2, P["rad_table"].reshape(-1, 2)) * P["rad_mask"]
    inp = torch.cat([xp2 * 2 - 1, enc2, feat, sh4((refl + 1) / 2) * P["rad_sh_mask"], nw], -1)
    hcur = torch.relu(inp @ P["rad_W0"].T + P["rad_b0"])
    hcur = torch.relu(hcur @ P["rad_W2"].T + P["rad_b2"])
    rgbs = torch.sigmoid(hcur @ P["rad_W4"].T + P["rad_b4"])
    # --- T2 / T3
    ri = fixed["ray_indices"]
    n_rays = fixed["n_rays"]
    trs = []
    for b, s in fixed["packed_info"].tolist():
        if s > 0:
            trs.append(torch.cat([torch.ones(1, dtype=x.dtype), torch.cumprod(1 - alphas[b:b + s], 0)[:-1]]))
    trans = torch.cat(trs) if trs else alphas[:0]
    w = trans * alphas
    comp = torch.zeros(n_rays, 3, dtype=x.dtype).index_add(0, ri, w[:, None] * rgbs)
    opac = torch.zeros(n_rays, 1, dtype=x.dtype).index_add(0, ri, w[:, None])
    loss = (comp - target_rgb).abs().mean()
    # .mean() over ALL samples (systems/intrinsic_avatar.py:235-237); invalid samples carry [0,0,1] -> 0 in the sum
    if valid.any():
        loss = loss + lambda_eik * ((sdf_grad[valid].norm(dim=-1) - 1.0) ** 2).sum() / max(sdf_grad.shape[0], 1)
    if target_mask is not None:
        op = opac[:, 0].clamp(1e-3, 1 - 1e-3)
        loss = loss + lambda_mask * torch.nn.functional.binary_cross_entropy(op, target_mask)
    if laplace is not None:
        loss = loss + lambda_curv * laplace.abs().mean()
    return loss, dict(comp_rgb=comp, opacity=opac, sdf_grad=sdf_grad, rgbs=rgbs, alphas=alphas, weights=w, feat=feat, xp2=xp2,
                      enc2=enc2)


# ----------------------------------------------------------------------------- PBR (fp64, differentiable)
def brdf_eval_t(n, wi, wo, alpha, albedo, metallic):
    """oracle/pbr_ref.brdf_eval in torch: (diff [F], spec [F,3]) incl. cosine."""
    NoL = (n * wo).sum(-1)
    NoV = (n * wi).sum(-1)
    h = wi + wo
    hl = h.norm(dim=-1, keepdim=True)
    h = h / hl.clamp_min(1e-30)
    NoH = (n * h).sum(-1)
    VoH = (wi * h).sum(-1).clamp_min(0.0)
    a2 = alpha ** 2
    dd = NoH ** 2 * (a2 - 1) + 1
    D = a2 / (math.pi * dd ** 2)
    G1 = lambda x: 2 * x / (x + torch.sqrt(a2 + (1 - a2) * x ** 2))      # noqa: E731
    common = D * G1(NoL) * G1(NoV) / (4 * NoV)
    F0 = 0.04 * (1 - metallic[:, None]) + albedo * metallic[:, None]
    Fr = F0 + (1 - F0) * ((1 - VoH) ** 5)[:, None]
    lit = NoL > 0
    diff = torch.where(lit, NoL / math.pi, torch.zeros_like(NoL))
    ok = lit & (NoV > 0) & (hl[:, 0] >= 1e-12)
    spec = torch.where(ok[:, None], common[:, None] * Fr, torch.zeros_like(Fr))
    return diff, spec


def env_eval_t(base, d):
    """bilinear equirect lookup (wrap in u, clamp in v) of base [H,W,3] at world directions d [F,3]."""
    H, W, _ = base.shape
    u = torch.atan2(d[:, 0], -d[:, 2]) / (2 * math.pi) + 0.5
    v = torch.acos(d[:, 1].clamp(-1, 1)) / math.pi
    fx, fy = u * W - 0.5, v * H - 0.5
    x0, y0 = torch.floor(fx), torch.floor(fy)
    ax, ay = (fx - x0)[:, None], (fy - y0)[:, None]
    x0, y0 = x0.long(), y0.long()
    x1, y1 = (x0 + 1) % W, (y0 + 1).clamp(0, H - 1)
    x0, y0 = x0 % W, y0.clamp(0, H - 1)
    return (1 - ax) * (1 - ay) * base[y0, x0] + ax * (1 - ay) * base[y0, x1] + (1 - ax) * ay * base[y1, x0] + ax * ay * base[y1, x1]


def pbr_uniform_light_t(n, albedo, rough, metal, view_dirs, wo, tr, base, R, inv_pdf):
    """pbr_uniform_light_forward (intrinsic_avatar.py:654-753) on given directions / transmittance, fp64."""
    cosm = (n * wo).sum(-1) > 1e-6
    t = tr.clamp(0, 1) * cosm
    diff, spec = brdf_eval_t(n, -view_dirs, wo, rough, albedo, metal)
    dw = torch.nn.functional.normalize(wo @ R, dim=-1)
    em = env_eval_t(base, dw)
    Li = em * t[:, None]
    w = (inv_pdf * cosm)[:, None]
    Ld, Ls = Li * diff[:, None] * w, Li * spec * w
    return ((1 - metal[:, None]) * albedo) * Ld + Ls, Ld, Ls


def pbr_light_t(n, albedo, rough, metal, view_dirs, wo, tr, ind_rgb, base, R, pdf):
    """pbr_light_forward (intrinsic_avatar.py:755-861) on given directions / transmittance / indirect radiance / light pdf
    (all no_grad constants there), fp64; differentiable in the materials, the normal and the environment image."""
    cosm = (n * wo).sum(-1) > 1e-6
    t = tr.clamp(0, 1)
    live = cosm & (t > 0)
    diff, spec = brdf_eval_t(n, -view_dirs, wo, rough, albedo, metal)
    dw = torch.nn.functional.normalize(wo @ R, dim=-1)
    em = torch.where(live[:, None], env_eval_t(base, dw), torch.zeros_like(dw))
    p = torch.where(live & (pdf > 0), pdf, torch.ones_like(pdf))
    Li = em * t[:, None] + (ind_rgb if ind_rgb is not None else 0.0)
    z = torch.zeros_like(Li)
    Ld = torch.where(cosm[:, None], Li * diff[:, None] / p[:, None], z)
    Ls = torch.where(cosm[:, None], Li * spec / p[:, None], z)
    return ((1 - metal[:, None]) * albedo) * Ld + Ls, Ld, Ls


def sg_image_t(axis, log_lambda, mu, base_res):
    """EnvironmentLightSG.generate_image in fp64: sum_k softplus(mu_k) exp(lambda_k (d . xi_k - 1)) on the equirect grid."""
    H, W = base_res, 2 * base_res
    v = (torch.arange(H, dtype=axis.dtype) + 0.5) / H
    u = (torch.arange(W, dtype=axis.dtype) + 0.5) / W
    th, ph = (v * math.pi)[:, None], ((u - 0.5) * 2 * math.pi)[None, :]
    d = torch.stack([torch.sin(th) * torch.sin(ph), torch.cos(th).expand(H, W), -torch.sin(th) * torch.cos(ph)], -1)
    xi = torch.nn.functional.normalize(axis, dim=-1)
    w = torch.exp(torch.exp(log_lambda) * (d.reshape(-1, 3) @ xi.T - 1.0))
    return (w @ torch.nn.functional.softplus(mu)).reshape(H, W, 3)


def shade_reference_phys(P, fixed, target_rgb, target_mask, background, lambda_phys=1.0, mode="uniform_light", **kw):
    """shade_reference + the physically based branch of the training step (BASELINE config 4), float64 autograd:
    material head on [radiance embedding | geometry feature] (models/intrinsic_avatar.py:1100-1113, pbr/material.py:31-51,
    LipshitzMLP network_utils.py:396-428), re-sampled weights w / count and attribute gathers (models/pbr/utils.py:137-206),
    uniform_light estimator on the GIVEN secondary rays (pbr_uniform_light_forward :654-753; directions, transmittance and
    inverse pdf are no_grad constants there too), Lo.scatter_ + accumulate (:1335-1342,:1420-1466), L1 on the image.
    mode 'light': pbr_light_forward (:755-861) instead, with fixed['light_pdf'] and (global illumination) fixed['sec_rgb'].
    fixed additionally carries: fg_src, fg_ray (int64 [F]), fg_counts [S], has_samples / has_bg (bool [n]), out_dirs, sec_tr,
    inv_pdf | light_pdf (, sec_rgb), env_R."""
    loss, r = shade_reference(P, fixed, target_rgb, target_mask, **kw)
    nrm = lambda v: v / v.norm(dim=-1, keepdim=True).clamp_min(1e-6)     # noqa: E731
    normal_smpl = nrm(r["sdf_grad"])
    inp = torch.cat([r["xp2"] * 2 - 1, r["enc2"], r["feat"]], -1)
    h = inp
    for i in range(3):
        Wm = P[f"mat_W{i}"]
        c = torch.nn.functional.softplus(P[f"mat_c{i}"])
        Wn = Wm * torch.clamp(c / Wm.abs().sum(dim=1), max=1.0)[:, None]
        h = h @ Wn.T + P[f"mat_b{i}"]
        h = torch.relu(h) if i < 2 else torch.sigmoid(h)
    albedo, rough, metal = h[:, :3] * 0.77 + 0.03, h[:, 3] * 0.9 + 0.09, h[:, 4]
    src, ray = fixed["fg_src"], fixed["fg_ray"]
    w_fg = r["weights"][src] / fixed["fg_counts"][src].double()
    if mode == "light":
        Lo, _, _ = pbr_light_t(normal_smpl[src], albedo[src], rough[src], metal[src], fixed["rays_d"][ray], fixed["out_dirs"],
                               fixed["sec_tr"], fixed.get("sec_rgb"), P["env_base"], fixed["env_R"], fixed["light_pdf"])
    else:
        Lo, _, _ = pbr_uniform_light_t(normal_smpl[src], albedo[src], rough[src], metal[src], fixed["rays_d"][ray], fixed["out_dirs"],
                                       fixed["sec_tr"], P["env_base"], fixed["env_R"], fixed["inv_pdf"])
    n_rays = fixed["n_rays"]
    img = torch.zeros(n_rays, 3, dtype=Lo.dtype).index_add(0, ray, w_fg[:, None] * Lo)
    T = (1.0 - r["opacity"][:, 0]) * fixed["has_bg"].double()
    img = img + T[:, None] * background[None]
    img = torch.where(fixed["has_samples"][:, None], img, background[None].expand(n_rays, 3))
    loss = loss + lambda_phys * (img - target_rgb).abs().mean()
    r.update(comp_rgb_phys=img, albedo=albedo, roughness=rough, metallic=metal)
    return loss, r
