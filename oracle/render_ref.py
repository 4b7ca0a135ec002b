"""CPU ORACLE (test infrastructure, NOT the product): numpy/C restatement of the reference's render_step
forward for BASELINE configs 1-2 (radiance + SDF geometry), composed from the oracle primitives.

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


def radiance(sc, pts_cano, feat, view_world, normal_world):
    xp = ((pts_cano - sc.rad_center) / sc.rad_scale + 0.5).astype(np.float32)
    enc = O.hashgrid_fwd(xp, sc.rad_params) * sc.rad_mask[None]
    x = -view_world
    dirs = 2 * (x * normal_world).sum(-1, keepdims=True) * normal_world - x       # reflect, models/utils.py:115
    sh = O.sh4(((dirs + 1) / 2).astype(np.float32)) * sc.rad_sh_mask[None]
    inp = np.concatenate([xp * 2 - 1, enc, feat, sh, normal_world], -1).astype(np.float32)
    return O.mlp_fwd(inp, sc.rad_W, sc.rad_b, "relu", "sigmoid")


def render_step(sc, rays_world, jitter=None, importance_sample=True):
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
