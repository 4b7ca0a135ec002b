"""CPU ORACLE (test infrastructure, NOT the product): numpy/C restatement of the reference's render_step
forward, composed from the oracle primitives: `render_step` = BASELINE configs 1-2 (radiance + SDF geometry),
`relight_step` = the same with the physically based branch (steps 6-8 of forward_, BASELINE configs 3 / 5 and the
forward half of config 4).

Follows IntrinsicAvatarModel.forward_ (models/intrinsic_avatar.py:950-1287), SNARFDeformer.deform
(models/deformers/snarf_deformer.py:187-261), VolumeSDF / VolumeRefDirRadiance forward
(models/rf/geometry.py:124-172, models/rf/radiance.py:111-135) and rendering_with_normals_sdf
(models/volrend.py:638-807).  Used by tests (end-to-end parity at 128x128) and by bench.py's
cpu_baseline leg ("port").
"""
import numpy as np

from . import oracle as O

INIT_BONES = np.array([0, 1, 2, 4, 5, 10, 11, 12, 15, 16, 17, 18, 19], np.int32)


class Scene:
    """plain-numpy bundle of everything the forward needs (all effective weights in REFERENCE column order)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def transform_rays_w2s(rays, w2s):
    o = rays[:, :3] @ w2s[:3, :3].T + w2s[None, :3, 3]
    d = rays[:, 3:6] @ w2s[:3, :3].T
    dist = np.linalg.norm(o, axis=-1, keepdims=True)
    return np.concatenate([o, d, dist - 1, dist + 1], -1).astype(np.float32)


def deform(sc, pts, with_grad=False, with_feature=False):
    """snarf_deformer.py:187-261 (eval mode, use_j_inv=false => c2w = blended bone rotation)."""
    P = pts.shape[0]
    x, Jinv, valid = O.fuse_broyden(pts[None], sc.voxel_J, sc.tfs, INIT_BONES, sc.offset_kernel, sc.scale_kernel)
    mask = O.filter(x, valid)[0]
    x = x[0]
    cand = x[mask]
    sdf_c, grad_c, feat_c = O.sdf_field(cand, sc.geo_center, sc.geo_scale, sc.geo_params, sc.geo_mask, sc.geo_W1, sc.geo_b1,
                                        sc.geo_W2, sc.geo_b2, with_grad=with_grad, with_feat=with_feature)
    sdf_all = np.full((P, 13), 1e5, np.float32)
    sdf_all[mask] = sdf_c
    idx = np.argmin(sdf_all, axis=1)            # first minimum
    sdf = sdf_all[np.arange(P), idx]
    x0 = x.copy()
    x0[~mask] = 0
    out = dict(pts_cano=x0[np.arange(P), idx], sdf=sdf, valid=mask.any(1))
    if with_grad:
        # fwd_tfs: blended bone rotation at the root = trilinear sample of voxel_J (linear in the weights)
        g_all = np.tile(np.array([0, 0, 1], np.float32), (P, 13, 1))
        gc_all = g_all.copy()
        gc_all[mask] = grad_c
        R = sample_voxel_J_rot(sc, cand)
        g_all[mask] = np.einsum("bij,bj->bi", R, grad_c)
        out["sdf_grad"] = g_all[np.arange(P), idx]
        out["sdf_grad_cano"] = gc_all[np.arange(P), idx]
    if with_feature:
        f_all = np.zeros((P, 13, 13), np.float32)
        f_all[mask] = feat_c
        out["feature"] = f_all[np.arange(P), idx]
    out["n_candidates"] = int(mask.sum())
    return out


def sample_voxel_J_rot(sc, xc):
    """3x3 part of the trilinearly interpolated voxel_J at canonical points (align_corners, inside the grid)."""
    vJ = sc.voxel_J[0]
    _, D, H, W = vJ.shape
    g = (xc + sc.offset_kernel.reshape(1, 3)) * sc.scale_kernel.reshape(1, 3)
    ix = (g[:, 0] + 1) / 2 * (W - 1)
    iy = (g[:, 1] + 1) / 2 * (H - 1)
    iz = (g[:, 2] + 1) / 2 * (D - 1)
    x0, y0, z0 = np.floor(ix).astype(int), np.floor(iy).astype(int), np.floor(iz).astype(int)
    out = np.zeros((xc.shape[0], 12), np.float32)
    for c in range(8):
        xx, yy, zz = x0 + (c & 1), y0 + ((c >> 1) & 1), z0 + ((c >> 2) & 1)
        w = (np.where(c & 1, ix - x0, x0 + 1 - ix) * np.where(c & 2, iy - y0, y0 + 1 - iy)
             * np.where(c & 4, iz - z0, z0 + 1 - iz)).astype(np.float32)
        inb = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
        v = vJ[:, np.clip(zz, 0, D - 1), np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].T
        out += np.where(inb[:, None], v * w[:, None], 0)
    return out.reshape(-1, 3, 4)[:, :, :3]


def _normalize(v, eps=1e-6):
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.maximum(n, eps)


def reflect(x, n):
    """models/utils.py:115-116."""
    return 2 * (x * n).sum(-1, keepdims=True) * n - x


def radiance(sc, pts_cano, feat, view_world, normal_world):
    xp = ((pts_cano - sc.rad_center) / sc.rad_scale + 0.5).astype(np.float32)
    enc = O.hashgrid_fwd(xp, sc.rad_params) * sc.rad_mask[None]
    dirs = reflect(-view_world, normal_world)
    sh = O.sh4(((dirs + 1) / 2).astype(np.float32)) * sc.rad_sh_mask[None]
    inp = np.concatenate([xp * 2 - 1, enc, feat, sh, normal_world], -1).astype(np.float32)
    return O.mlp_fwd(inp, sc.rad_W, sc.rad_b, "relu", "sigmoid")


def render_step(sc, rays_world, jitter=None, importance_sample=True, _sampling_only=False):
    rays = transform_rays_w2s(rays_world, sc.w2s)
    n = rays.shape[0]
    ro, rd, far = rays[:, :3], rays[:, 3:6], rays[:, 7]
    near_p = np.zeros(n, np.float32)
    if jitter is not None:
        near_p = near_p + jitter * np.float32(sc.step)
    far_p = np.full(n, 1e10, np.float32)
    tr = O.traverse_grids(ro, rd, sc.binaries, sc.aabb, near_p, far_p, sc.step)
    iv = tr["intervals"]
    vals, il, ir, ridx, pi = iv["vals"], iv["is_left"], iv["is_right"], iv["ray_indices"], iv["packed_info"].astype(np.int32)
    stats = dict(n_edges0=len(vals), n_samples0=len(tr["samples"]["vals"]))
    if importance_sample and len(vals) > 0:
        for it in range(2):
            if it == 0:
                pts = ro[ridx] + rd[ridx] * vals[:, None]
                sdf = deform(sc, pts)["sdf"]
                sdf_m = np.full_like(sdf, 1e10)
                sdf_m[il] = np.minimum(sdf[il], sdf[ir])
                alphas = O.laplace_alpha(sdf_m, np.full_like(sdf, sc.step), sc.beta)
            else:
                ts, te = vals[il], vals[ir]
                pts = ro[ridx[il]] + rd[ridx[il]] * ((ts + te) / np.float32(2.0))[:, None]
                sdf_c = deform(sc, pts)["sdf"]
                sdf = np.full_like(vals, 1e10)
                sdf[il] = sdf_c
                dists = np.zeros_like(vals)
                dists[il] = te - ts
                alphas = O.laplace_alpha(sdf, dists, sc.beta)
            w, _ = O.render_weight_from_alpha(alphas, pi)
            rpi, rv, rdi, rl, rr, rs, fg = O.ray_resampling_merge(pi, vals, il, ir, w, 16)
            ridx = O.unpack_info(rpi, len(rv))[fg]
            vals, il, ir = rv[fg], rl[fg], rr[fg]
            pi = O.pack_info(ridx, n)
    ts, te, rix = vals[il], vals[ir], ridx[il]
    pinfo = O.pack_info(rix, n)
    if _sampling_only:
        return ro, rd, far, ts, te, rix, pinfo, stats
    pts = ro[rix] + rd[rix] * ((ts + te) / np.float32(2.0))[:, None]
    d = deform(sc, pts, with_grad=True, with_feature=True)
    R = sc.w2s[:3, :3]
    normal_world = _normalize(d["sdf_grad"] @ R)
    view_world = _normalize(rd[rix] @ R)
    alphas = O.laplace_alpha(d["sdf"], te - ts, sc.beta)
    rgbs = radiance(sc, d["pts_cano"], d["feature"], view_world.astype(np.float32), normal_world.astype(np.float32))
    w, tr_ = O.render_weight_from_alpha(alphas, pinfo)
    col = O.accumulate_along_rays(w, rgbs, rix, n)
    nrm = O.accumulate_along_rays(w, normal_world.astype(np.float32), rix, n)
    opa = O.accumulate_along_rays(w, None, rix, n)
    dep = O.accumulate_along_rays(w, ((ts + te) / np.float32(2.0))[:, None], rix, n)
    dep = dep + (1 - opa) * far[:, None]
    stats.update(n_samples=len(ts), n_candidates=d["n_candidates"])
    return dict(comp_rgb=col, comp_normal=nrm, opacity=opa, depth=dep, weights=w, alphas=alphas, rgbs=rgbs, sdf=d["sdf"],
                sdf_grad=d["sdf_grad"], t_starts=ts, t_ends=te, ray_indices=rix, packed_info=pinfo, stats=stats)


# =============================================================================================================
# steps 6-8 of forward_ (enable_phys): materials, volume-interaction re-sampling, secondary rays, PBR estimator
# =============================================================================================================
MAT_SCALE = np.array([0.77, 0.77, 0.77, 0.9, 1.0], np.float32)       # models/pbr/material.py:24-29 defaults
MAT_BIAS = np.array([0.03, 0.03, 0.03, 0.09, 0.0], np.float32)


def material(sc, pts_cano, feat):
    """VolumeMaterial.forward on material_feature = hybrid (models/intrinsic_avatar.py:1100-1113, models/pbr/material.py:31-51):
    [radiance xyz embedding (35) | geometry feature (13)] -> LipshitzMLP -> sigmoid -> affine.  sc.mat_W / sc.mat_b are the
    Lipschitz-normalised weights (network_utils.py:396-403) in the reference's column order."""
    xp = ((pts_cano - sc.rad_center) / sc.rad_scale + 0.5).astype(np.float32)
    enc = O.hashgrid_fwd(xp, sc.rad_params) * sc.rad_mask[None]
    inp = np.concatenate([xp * 2 - 1, enc, feat], -1).astype(np.float32)
    return O.mlp_fwd(inp, sc.mat_W, sc.mat_b, "relu", "sigmoid") * MAT_SCALE[None] + MAT_BIAS[None]


def shade_points(sc, ro, rd, rix, ts, te, with_materials=False):
    """rgb_normal[_mats]_alpha_fn (models/intrinsic_avatar.py:1032-1156, eval): deformer + SDF(grad, feature) -> normals,
    alpha, radiance (, materials) at the interval mid-points."""
    pts = ro[rix] + rd[rix] * ((ts + te) / np.float32(2.0))[:, None]
    d = deform(sc, pts, with_grad=True, with_feature=True)
    R = sc.w2s[:3, :3]
    normal_smpl = _normalize(d["sdf_grad"]).astype(np.float32)                  # F.normalize(sdf_grad, eps=1e-6)
    normal_world = _normalize(d["sdf_grad"] @ R).astype(np.float32)             # transform_dirs_s2w
    view_world = _normalize(rd[rix] @ R).astype(np.float32)
    alphas = O.laplace_alpha(d["sdf"], te - ts, sc.beta)
    rgbs = radiance(sc, d["pts_cano"], d["feature"], view_world, normal_world)
    out = dict(d, normal_smpl=normal_smpl, normal_world=normal_world, alphas=alphas, rgbs=rgbs)
    if with_materials:
        out["materials"] = material(sc, d["pts_cano"], d["feature"])
    return out


def compute_indirect_radiance(sc, ro, rd, near=0.0, far=1.5, n_secondary=64):
    """models/intrinsic_avatar.py:396-545 (eval): march [near, far] through the occupancy grid with step (far-near)/63,
    SDF at the interval STARTS (coarse_alpha_sdf_fn :399-428), zero-crossing re-sampling to 4 intervals (K4, :490-505), keep
    the foreground intervals, shade them (rgb_alpha_fn :430-456) and composite with `rendering` (volrend.py:19-194).
    returns (1 - acc [M,1], rgb [M,3], stats)."""
    M = ro.shape[0]
    step = np.float32((far - near) / (n_secondary - 1))
    tr = O.traverse_grids(ro, rd, sc.binaries, sc.aabb, np.full(M, near, np.float32), np.full(M, far, np.float32), step)
    iv = tr["intervals"]
    ts, te, rix = iv["vals"][iv["is_left"]], iv["vals"][iv["is_right"]], tr["samples"]["ray_indices"]
    stats = dict(n_secondary_samples=len(ts), n_secondary_fg=0)
    acc = np.zeros((M, 1), np.float32)
    rgb = np.zeros((M, 3), np.float32)
    if len(ts) == 0:
        return 1.0 - acc, rgb, stats
    sdf = deform(sc, ro[rix] + rd[rix] * ts[:, None])["sdf"]
    alphas = O.laplace_alpha(sdf, te - ts, sc.beta)
    pinfo = O.pack_info(rix, M)
    rpi, rs, re, is_fg = O.ray_resampling_sdf_fine(pinfo, ts[:, None], te[:, None], alphas, sdf, 4)
    rri = O.unpack_info(rpi, rs.shape[0])
    rix, ts, te = rri[is_fg], rs[is_fg, 0], re[is_fg, 0]
    stats["n_secondary_fg"] = len(ts)
    if len(ts) == 0:
        return 1.0 - acc, rgb, stats
    sh = shade_points(sc, ro, rd, rix, ts, te)
    pinfo = O.pack_info(rix, M)
    w, _ = O.render_weight_from_alpha(sh["alphas"], pinfo)
    acc = O.accumulate_along_rays(w, None, rix, M)
    rgb = O.accumulate_along_rays(w, sh["rgbs"], rix, M)
    return 1.0 - acc, rgb, stats


def sample_volume_interaction(ro, rd, rix, ts, te, n_rays, spp, transmittance_map, extras):
    """models/pbr/utils.py:70-229: K1 re-sampling of the un-normalised weight CDF (+ background bin), fg / bg split by the
    1e4 offset marker, re-sampled weights = w / count (fg) and (1 - acc) / count (bg), attribute gathers."""
    weights, sdfs = extras["weights"], extras["sdf"]
    pinfo = O.pack_info(rix, n_rays)
    rpi, mid, offs, sidx, fg_cnt, bg_cnt, surf = O.ray_resampling(pinfo, ts[:, None], te[:, None], weights, sdfs, spp)
    fg_idx = np.nonzero(offs[:, 0] < 1e4)[0]
    bg_idx = np.nonzero(offs[:, 0] >= 1e4)[0]
    rri = O.unpack_info(rpi, mid.shape[0])
    fg_rri, bg_rri = rri[fg_idx], rri[bg_idx]
    fg_s = sidx[fg_idx]
    rw = np.zeros(mid.shape[0], np.float32)
    ex = {}
    if len(fg_s) > 0:
        rw[fg_idx] = weights[fg_s] / fg_cnt[fg_s].astype(np.float32)
        rw[bg_idx] = transmittance_map[bg_rri, 0] / bg_cnt[bg_rri].astype(np.float32)
        t = mid[fg_idx]
        ex = dict(sdf=sdfs[fg_s], alphas=extras["alphas"][fg_s], dists=(te - ts)[:, None][fg_s],
                  positions=(ro[fg_rri] + rd[fg_rri] * t).astype(np.float32), normals=extras["normals"][fg_s],
                  albedo=extras["albedo"][fg_s], roughness=extras["roughness"][fg_s], metallic=extras["metallic"][fg_s],
                  t_dirs=rd[fg_rri])
    return rpi, rri, rw, fg_idx, bg_idx, ex, dict(sampled_indices=sidx, fg_counts=fg_cnt, bg_counts=bg_cnt, midpoints=mid,
                                                   offsets=offs, surface_idx=surf)


def light_shuffle(n_rays, spp, rpi, fg_idx, shuffle_u):
    """models/intrinsic_avatar.py:1356-1378: per-ray permutation of [0, spp) = argsort of uniforms (explicit here; ties --
    which torch.argsort leaves unspecified -- by index), unpack_data mask -> pack_data -> restrict to fg re-samples."""
    col = np.argsort(shuffle_u, axis=-1, kind="stable")
    has = rpi[:, 1] > 0
    return col[has].reshape(-1)[fg_idx]


def _light_estimators(sc, Pb, ex, pmf, R, F_, n, spp, rpi, rri, rw, fg_idx, light_u, shuffle_u, per_point, render_mode,
                      global_illumination, stats):
    """render_mode light / uniform_light of relight_step: directions, cosine mask, secondary rays, estimator."""
    vis_map = None
    shuffled = None
    if per_point:                                                                    # training branch (:777-781)
        u = light_u[:F_]
        dirs_world = Pb.envlight_sample(pmf, F_, u[:, 0].astype(np.float64), u[:, 1].astype(np.float64), u[:, 2].astype(np.float64))
        out_dirs = _normalize(dirs_world @ R.T).astype(np.float32)                   # transform_dirs_w2s
    else:
        if render_mode == "light":
            dirs_world = Pb.envlight_sample(pmf, spp, light_u[:, 0].astype(np.float64), light_u[:, 1].astype(np.float64),
                                            light_u[:, 2].astype(np.float64))
            dirs_smpl = _normalize(dirs_world @ R.T).astype(np.float32)             # transform_dirs_w2s
            inv_pdf_all = None
        else:                                                                        # uniform_light (:654-753, :1390-1401)
            assert render_mode == "uniform_light" and spp == 512
            dirs_smpl, inv_pdf_all = Pb.uniform_sphere_stratified(16, 32, light_u[:, :2])
        shuffled = light_shuffle(n, spp, rpi, fg_idx, shuffle_u)
        out_dirs = dirs_smpl[shuffled]
    cos_mask = (ex["normals"] * out_dirs).sum(-1) > 1e-6
    sec_tr = np.zeros((F_, 1), np.float32)
    sec_rgb = np.zeros((F_, 3), np.float32)
    stats["n_secondary"] = int(cos_mask.sum())
    if stats["n_secondary"] > 0:
        t_, c_, st2 = compute_indirect_radiance(sc, np.ascontiguousarray(ex["positions"][cos_mask]),
                                                np.ascontiguousarray(out_dirs[cos_mask]))
        sec_tr[cos_mask], sec_rgb[cos_mask] = np.clip(t_, 0.0, 1.0), c_
        stats.update(st2)
    if render_mode == "light":
        fg_Lo, fg_Ld, fg_Ls = Pb.pbr_light_shade(ex["normals"], ex["albedo"], ex["roughness"][:, 0], ex["metallic"][:, 0],
                                                 ex["t_dirs"], out_dirs, sec_tr[:, 0],
                                                 sec_rgb if global_illumination else None, sc.env_base, pmf, R)
    else:
        fg_Lo, fg_Ld, fg_Ls, fg_vis = Pb.pbr_uniform_light_shade(
            ex["normals"], ex["albedo"], ex["roughness"][:, 0], ex["metallic"][:, 0], ex["t_dirs"], out_dirs, sec_tr[:, 0],
            sec_rgb if global_illumination else None, sc.env_base, R, inv_pdf_all[shuffled][:, 0])
        vis = np.zeros((len(rri), 3), np.float32)
        vis[fg_idx] = fg_vis
        vis_map = O.accumulate_along_rays(rw, vis, rri, n).mean(-1, keepdims=True)      # :1414-1419
    return fg_Lo, cos_mask, sec_tr, sec_rgb, out_dirs, shuffled, fg_Ld, fg_Ls, vis_map


def relight_step(sc, rays_world, spp=16, seed=0, light_u=None, shuffle_u=None, jitter=None, global_illumination=False,
                 background_color=(1.0, 1.0, 1.0), importance_sample=True, render_mode="light", light_sampling="shared",
                 scatter_u=None):
    """forward_ with enable_phys, render_mode = light, eval form (models/intrinsic_avatar.py:950-1651): steps 1-4 as
    render_step, step 5 = rendering_with_normals_mats_sdf (volrend.py:810-1020), steps 6-8 = :1288-1470.
    light_u [spp,3] (emitter.sample uniforms) and shuffle_u [n_rays,spp] are explicit (drawn from `seed` when None).
    light_sampling (render_mode 'light'): 'shared' = the eval branch of pbr_light_forward (:782-789: one set of spp directions
    per frame, permuted per ray); 'per_point' = its `self.training` branch (:777-781): emitter.sample(F) -- an independent
    direction per foreground re-sample, light_u [>= F, 3], no shuffle.
    render_mode 'mats' / 'mis' (pbr_mats_forward :863-948, pbr_mis_forward :547-652): scatter_u [>= F, 6] = the uniforms of
    scatterer.sample (columns 0..2) and, for mis, of emitter.sample(F) (columns 3..5); every fg re-sample traces its rays.
    sc additionally carries mat_W, mat_b, env_base [H,W,3]."""
    from . import pbr_ref as Pb
    rng = np.random.default_rng(seed)
    n = rays_world.shape[0]
    per_point = render_mode == "light" and light_sampling == "per_point"
    if light_u is None:
        light_u = rng.random(((n * spp) if per_point else spp, 3), dtype=np.float32)
    if shuffle_u is None and not per_point:
        shuffle_u = rng.random((n, spp), dtype=np.float32)
    bgc = np.asarray(background_color, np.float32)
    base = render_step(sc, rays_world, jitter=jitter, importance_sample=importance_sample, _sampling_only=True)
    ro, rd, far, ts, te, rix, pinfo, stats = base
    sh = shade_points(sc, ro, rd, rix, ts, te, with_materials=True)
    w, trn = O.render_weight_from_alpha(sh["alphas"], pinfo)
    acc = lambda v: O.accumulate_along_rays(w, v, rix, n)      # noqa: E731
    mats = sh["materials"]
    out = dict(comp_rgb=acc(sh["rgbs"]), comp_normal=acc(sh["normal_world"]), albedo=acc(np.ascontiguousarray(mats[:, :3])),
               roughness=acc(np.ascontiguousarray(mats[:, 3:4])), metallic=acc(np.ascontiguousarray(mats[:, 4:5])),
               opacity=acc(None), weights=w, alphas=sh["alphas"], sdf=sh["sdf"], sdf_grad=sh["sdf_grad"], t_starts=ts, t_ends=te,
               ray_indices=rix, packed_info=pinfo)
    out["depth"] = acc(((ts + te) / np.float32(2.0))[:, None]) + (1 - out["opacity"]) * far[:, None]
    rgb_phys = np.tile(bgc[None], (n, 1)).astype(np.float32)
    demod_phys = rgb_phys.copy()
    if render_mode == "uniform_light":
        out["visibility"] = np.zeros((n, 1), np.float32)                                      # :1266-1267
    stats.update(n_samples=len(ts), n_resampled=0, n_fg=0, n_secondary=0)
    if len(rix) > 0:
        extras = dict(weights=w, sdf=sh["sdf"], alphas=sh["alphas"], normals=sh["normal_smpl"], albedo=mats[:, :3],
                      roughness=mats[:, 3:4], metallic=mats[:, 4:5])
        rpi, rri, rw, fg_idx, bg_idx, ex, k1 = sample_volume_interaction(ro, rd, rix, ts, te, n, spp, 1.0 - out["opacity"], extras)
        stats.update(n_resampled=len(rri), n_fg=len(fg_idx))
        out.update(resampled_packed_info=rpi, resampled_weights=rw, fg_indices=fg_idx, bg_indices=bg_idx, k1=k1)
        if len(fg_idx) > 0:
            F_ = len(fg_idx)
            R = sc.w2s[:3, :3]
            pmf = Pb.envlight_pmf(sc.env_base)
            shuffled = None
            if render_mode in ("mats", "mis"):
                assert scatter_u is not None and scatter_u.shape[0] >= F_
                rough, metal = ex["roughness"][:, 0], ex["metallic"][:, 0]
                sc_dirs = Pb.brdf_sample(ex["normals"], -ex["t_dirs"], rough, scatter_u[:F_, :3].astype(np.float64))
                if render_mode == "mis":
                    u = scatter_u[:F_, 3:6].astype(np.float64)
                    li_dirs = _normalize(Pb.envlight_sample(pmf, F_, u[:, 0], u[:, 1], u[:, 2]) @ R.T).astype(np.float32)
                    out_dirs = np.concatenate([sc_dirs, li_dirs], 0)
                    pos2 = np.concatenate([ex["positions"], ex["positions"]], 0)
                else:
                    out_dirs, pos2 = sc_dirs, ex["positions"]
                stats["n_secondary"] = int(out_dirs.shape[0])
                t_, c_, st2 = compute_indirect_radiance(sc, np.ascontiguousarray(pos2), np.ascontiguousarray(out_dirs))
                stats.update(st2)
                sec_tr, sec_rgb = t_, c_
                ind = sec_rgb if global_illumination else None
                if render_mode == "mis":
                    fg_Lo, fg_Ld, fg_Ls = Pb.pbr_mis_shade(ex["normals"], ex["albedo"], rough, metal, ex["t_dirs"], sc_dirs, li_dirs,
                                                           sec_tr[:, 0], ind, sc.env_base, pmf, R)
                else:
                    fg_Lo, fg_Ld, fg_Ls = Pb.pbr_mats_shade(ex["normals"], ex["albedo"], rough, metal, ex["t_dirs"], sc_dirs, sec_tr[:, 0],
                                                            ind, sc.env_base, R)
                cos_mask = np.ones(out_dirs.shape[0], bool)
            else:
                fg_Lo, cos_mask, sec_tr, sec_rgb, out_dirs, shuffled, fg_Ld, fg_Ls, vis_map = _light_estimators(
                    sc, Pb, ex, pmf, R, F_, n, spp, rpi, rri, rw, fg_idx, light_u, shuffle_u, per_point, render_mode, global_illumination, stats)
                if vis_map is not None:
                    out["visibility"] = vis_map
            Lo = np.zeros((len(rri), 3), np.float32)
            Lo[bg_idx] = bgc[None]                                                           # :1335-1342
            Lo[fg_idx] = fg_Lo
            rgb_phys = O.accumulate_along_rays(rw, Lo, rri, n)
            Lo_demod = Lo.copy()                                                             # :1336,1421-1423
            Lo_demod[fg_idx] = fg_Ld + fg_Ls
            demod_phys = O.accumulate_along_rays(rw, Lo_demod, rri, n)
            out.update(fg_Lo=fg_Lo, fg_Lo_diff=fg_Ld, fg_Lo_spec=fg_Ls, secondary_tr=sec_tr, secondary_rgb=sec_rgb,
                       out_dirs=out_dirs, cos_mask=cos_mask, fg_extras=ex, shuffled=shuffled)
        rgb_phys[rpi[:, 1] <= 0] = bgc[None]                                                 # :1452-1466
        demod_phys[rpi[:, 1] <= 0] = bgc[None]
    out.update(comp_rgb_phys=rgb_phys, comp_demod_phys=demod_phys, stats=stats)
    return out


def rgb_to_srgb(f):
    f = np.clip(f, 0.0, 1.0)
    return np.where(f <= 0.0031308, f * 12.92, np.power(np.maximum(f, 0.0031308), 1.0 / 2.4) * 1.055 - 0.055).astype(np.float32)


def forward_output_dict(o, background_color, render_mode="light"):
    """the dict IntrinsicAvatarModel.forward_ returns in eval mode with enable_phys (models/intrinsic_avatar.py:1492-1651),
    from a relight_step result: linear maps, the constant-background dict (`*_bg`) and the composited sRGB dict (`*_full`)."""
    bgc = np.asarray(background_color, np.float32)
    acc = o["opacity"]
    n = acc.shape[0]
    out = dict(comp_rgb=o["comp_rgb"], comp_normal=o["comp_normal"], opacity=acc, depth=o["depth"], rays_valid=acc > 0,
               rays_valid_phys=acc > 0, num_samples=np.array([len(o["t_starts"])], np.int32), comp_rgb_phys=o["comp_rgb_phys"],
               comp_demod_phys=o["comp_demod_phys"], comp_albedo=o["albedo"], comp_metallic=o["metallic"], comp_roughness=o["roughness"])
    if render_mode == "uniform_light":
        out["visibility"] = o["visibility"]
    bgm = np.full((n, 1), bgc.mean(), np.float32)
    out_bg = dict(comp_rgb=np.tile(bgc[None], (n, 1)), num_samples=np.zeros(1, np.int32), rays_valid=np.zeros((n, 1), bool),
                  rays_valid_phys=np.zeros((n, 1), bool), comp_albedo=np.zeros((n, 3), np.float32), comp_metallic=bgm, comp_roughness=bgm)
    T = 1.0 - acc
    out_full = dict(comp_rgb=np.clip(rgb_to_srgb(out["comp_rgb"] + out_bg["comp_rgb"] * T), 0, 1), num_samples=out["num_samples"],
                    rays_valid=out["rays_valid"], rays_valid_phys=out["rays_valid_phys"],
                    comp_rgb_phys=np.clip(rgb_to_srgb(out["comp_rgb_phys"]), 0, 1), comp_demod_phys=np.clip(rgb_to_srgb(out["comp_demod_phys"]), 0, 1),
                    comp_albedo=out["comp_albedo"] + out_bg["comp_albedo"] * T, comp_metallic=out["comp_metallic"] + out_bg["comp_metallic"] * T,
                    comp_roughness=out["comp_roughness"] + out_bg["comp_roughness"] * T)
    return {**out, **{k + "_bg": v for k, v in out_bg.items()}, **{k + "_full": v for k, v in out_full.items()}}
