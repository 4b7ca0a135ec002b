"""numpy front-end of the CPU ORACLE (oracle/libia_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the shipped package.  Each function
mirrors the reference operator of the same name and returns numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libia_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("ia_oracle.c", "ia_oracle_field.c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ia_ref_resample_packed_info.restype = C.c_int64
        _lib.ia_ref_hashgrid_offsets.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _u8(a):
    return np.ascontiguousarray(a).astype(np.uint8, copy=False)


# ----------------------------------------------------------------------------- T1
def traverse_grids(rays_o, rays_d, binaries, aabb, near_planes, far_planes, step_size, cone_angle=0.0):
    """nerfacc.traverse_grids for ONE grid level. binaries [rx,ry,rz] bool, aabb [6].

    Returns dict(intervals=..., samples=..., termination_planes=...)."""
    L = lib()
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    n = rays_o.shape[0]
    binaries = _u8(binaries.reshape(binaries.shape[-3:]))
    res = _i32(binaries.shape)
    aabb = _f32(np.asarray(aabb).reshape(6))
    near_planes, far_planes = _f32(near_planes), _f32(far_planes)
    iv_cnt = np.zeros(n, np.int64)
    sm_cnt = np.zeros(n, np.int64)
    L.ia_ref_traverse_grids_count(
        C.c_int64(n), _p(rays_o), _p(rays_d), _p(res), _p(binaries), _p(aabb), _p(near_planes), _p(far_planes),
        C.c_float(step_size), C.c_float(cone_angle), _p(iv_cnt), _p(sm_cnt))
    iv_start = np.cumsum(iv_cnt) - iv_cnt
    sm_start = np.cumsum(sm_cnt) - sm_cnt
    E, S = int(iv_cnt.sum()), int(sm_cnt.sum())
    iv_vals = np.zeros(E, np.float32)
    iv_l = np.zeros(E, np.uint8)
    iv_r = np.zeros(E, np.uint8)
    iv_ray = np.zeros(E, np.int64)
    sm_vals = np.zeros(S, np.float32)
    sm_ray = np.zeros(S, np.int64)
    term = np.zeros(n, np.float32)
    L.ia_ref_traverse_grids_fill(
        C.c_int64(n), _p(rays_o), _p(rays_d), _p(res), _p(binaries), _p(aabb), _p(near_planes), _p(far_planes),
        C.c_float(step_size), C.c_float(cone_angle), _p(iv_start), _p(sm_start),
        _p(iv_vals), _p(iv_l), _p(iv_r), _p(iv_ray), _p(sm_vals), _p(sm_ray), _p(term))
    return dict(
        intervals=dict(vals=iv_vals, is_left=iv_l.astype(bool), is_right=iv_r.astype(bool), ray_indices=iv_ray,
                       packed_info=np.stack([iv_start, iv_cnt], -1)),
        samples=dict(vals=sm_vals, ray_indices=sm_ray, packed_info=np.stack([sm_start, sm_cnt], -1)),
        termination_planes=term,
    )


# ----------------------------------------------------------------------------- T2 / T3
def render_weight_from_alpha(alphas, packed_info):
    alphas = _f32(alphas)
    packed_info = _i64(packed_info)
    w = np.zeros_like(alphas)
    t = np.zeros_like(alphas)
    lib().ia_ref_render_weight_from_alpha(C.c_int64(packed_info.shape[0]), _p(packed_info), _p(alphas), _p(w), _p(t))
    return w, t


def render_weight_from_alpha_bwd(alphas, packed_info, weights, trans, g_weights, g_trans=None):
    alphas, weights, trans = _f32(alphas), _f32(weights), _f32(trans)
    packed_info = _i64(packed_info)
    gw = _f32(g_weights) if g_weights is not None else None
    gt = _f32(g_trans) if g_trans is not None else None
    ga = np.zeros_like(alphas)
    lib().ia_ref_render_weight_from_alpha_bwd(C.c_int64(packed_info.shape[0]), _p(packed_info), _p(alphas),
                                               _p(weights), _p(trans), _p(gw), _p(gt), _p(ga))
    return ga


def accumulate_along_rays(weights, values, ray_indices, n_rays):
    weights = _f32(weights)
    ray_indices = _i64(ray_indices)
    dim = 1 if values is None else values.shape[-1]
    values = None if values is None else _f32(values)
    out = np.zeros((n_rays, dim), np.float32)
    lib().ia_ref_accumulate_along_rays(C.c_int64(weights.shape[0]), C.c_int64(n_rays), C.c_int(dim), _p(weights),
                                       _p(values), _p(ray_indices), _p(out))
    return out


# ----------------------------------------------------------------------------- pack / unpack
def pack_info(ray_indices, n_rays):
    ray_indices = _i64(ray_indices)
    out = np.zeros((n_rays, 2), np.int32)
    lib().ia_ref_pack_info(C.c_int64(ray_indices.shape[0]), _p(ray_indices), C.c_int64(n_rays), _p(out))
    return out


def unpack_info(packed_info, n_samples):
    packed_info = _i32(packed_info)
    out = np.zeros(n_samples, np.int64)
    lib().ia_ref_unpack_info(C.c_int64(packed_info.shape[0]), _p(packed_info), _p(out))
    return out


def unpack_info_to_mask(packed_info, n_samples):
    packed_info = _i32(packed_info)
    out = np.zeros((packed_info.shape[0], n_samples), np.uint8)
    lib().ia_ref_unpack_info_to_mask(C.c_int64(packed_info.shape[0]), _p(packed_info), C.c_int(n_samples), _p(out))
    return out.astype(bool)


def unpack_data(packed_info, data, n_samples):
    packed_info = _i32(packed_info)
    data = _f32(data)
    out = np.zeros((packed_info.shape[0], n_samples, data.shape[1]), np.float32)
    lib().ia_ref_unpack_data(C.c_int64(packed_info.shape[0]), _p(packed_info), C.c_int(data.shape[1]), _p(data),
                             C.c_int(n_samples), _p(out))
    return out


def _resample_info(packed_info, n, add_steps):
    out = np.zeros_like(packed_info)
    total = lib().ia_ref_resample_packed_info(C.c_int64(packed_info.shape[0]), _p(packed_info), C.c_int(n),
                                              C.c_int(add_steps), _p(out))
    return out, int(total)


# ----------------------------------------------------------------------------- K1..K4
def ray_resampling(packed_info, t_starts, t_ends, weights, sdfs, n):
    packed_info = _i32(packed_info)
    st, en, w, sd = _f32(t_starts).reshape(-1), _f32(t_ends).reshape(-1), _f32(weights), _f32(sdfs)
    rpi, T = _resample_info(packed_info, n, 0)
    ts = np.zeros(T, np.float32)
    offs = np.zeros(T, np.float32)
    idxs = np.zeros(T, np.int64)
    fg = np.zeros(w.shape[0], np.int32)
    bg = np.zeros(packed_info.shape[0], np.int32)
    surf = -np.ones(packed_info.shape[0], np.int64)
    lib().ia_ref_ray_resampling(C.c_int64(packed_info.shape[0]), _p(packed_info), _p(st), _p(en), _p(w), _p(sd),
                                _p(rpi), _p(ts), _p(offs), _p(surf), _p(idxs), _p(fg), _p(bg))
    return rpi, ts[:, None], offs[:, None], idxs, fg, bg, surf


def ray_resampling_merge(packed_info, vals, is_left, is_right, weights, n):
    packed_info = _i32(packed_info)
    vals, w = _f32(vals), _f32(weights)
    il, ir = _u8(is_left), _u8(is_right)
    rpi, T = _resample_info(packed_info, n, 1)
    ov = np.zeros(T, np.float32)
    od = np.zeros(T, np.float32)
    ol = np.zeros(T, np.uint8)
    orr = np.zeros(T, np.uint8)
    ors = np.zeros(T, np.uint8)
    fg = np.zeros(T, np.uint8)
    lib().ia_ref_ray_resampling_merge(C.c_int64(packed_info.shape[0]), _p(packed_info), _p(vals), _p(il), _p(ir),
                                      _p(w), _p(rpi), _p(ov), _p(od), _p(ol), _p(orr), _p(ors), _p(fg))
    return rpi, ov, od, ol.astype(bool), orr.astype(bool), ors.astype(bool), fg.astype(bool)


def ray_resampling_fine(packed_info, t_starts, t_ends, weights, n):
    packed_info = _i32(packed_info)
    st, en, w = _f32(t_starts).reshape(-1), _f32(t_ends).reshape(-1), _f32(weights)
    rpi, T = _resample_info(packed_info, n, 0)
    os_ = np.zeros(T, np.float32)
    oe = np.zeros(T, np.float32)
    fg = np.zeros(T, np.uint8)
    lib().ia_ref_ray_resampling_fine(C.c_int64(packed_info.shape[0]), _p(packed_info), _p(st), _p(en), _p(w),
                                     _p(rpi), _p(os_), _p(oe), _p(fg))
    return rpi, os_[:, None], oe[:, None], fg.astype(bool)


def ray_resampling_sdf_fine(packed_info, t_starts, t_ends, alphas, sdfs, n):
    packed_info = _i32(packed_info)
    st, en, al, sd = _f32(t_starts).reshape(-1), _f32(t_ends).reshape(-1), _f32(alphas), _f32(sdfs)
    rpi, T = _resample_info(packed_info, n, 0)
    os_ = np.zeros(T, np.float32)
    oe = np.zeros(T, np.float32)
    fg = np.zeros(T, np.uint8)
    lib().ia_ref_ray_resampling_sdf_fine(C.c_int64(packed_info.shape[0]), _p(packed_info), _p(st), _p(en), _p(al),
                                         _p(sd), _p(rpi), _p(os_), _p(oe), _p(fg))
    return rpi, os_[:, None], oe[:, None], fg.astype(bool)


# ----------------------------------------------------------------------------- fast-SNARF
def precompute(voxel_w, tfs, offset, scale):
    voxel_w, tfs = _f32(voxel_w), _f32(tfs)
    _, _, D, H, W = voxel_w.shape
    B = tfs.shape[0]
    offset, scale = _f32(np.asarray(offset).reshape(3)), _f32(np.asarray(scale).reshape(3))
    vd = np.zeros((B, 3, D, H, W), np.float32)
    vJ = np.zeros((B, 12, D, H, W), np.float32)
    lib().ia_ref_precompute(C.c_int(B), C.c_int(D), C.c_int(H), C.c_int(W), _p(voxel_w), _p(tfs), _p(offset),
                            _p(scale), _p(vd), _p(vJ))
    return vd, vJ


def fuse_broyden(xd_tgt, voxel_J, tfs, bone_ids, offset, scale, cvg=1e-5, dvg=1e-1):
    xd_tgt, voxel_J, tfs = _f32(xd_tgt), _f32(voxel_J), _f32(tfs)
    bone_ids = _i32(bone_ids)
    B, N, _ = xd_tgt.shape
    _, _, D, H, W = voxel_J.shape
    I = bone_ids.shape[0]
    offset, scale = _f32(np.asarray(offset).reshape(3)), _f32(np.asarray(scale).reshape(3))
    x = np.zeros((B, N, I, 3), np.float32)
    Ji = np.zeros((B, N, I, 3, 3), np.float32)
    valid = np.zeros((B, N, I), np.uint8)
    lib().ia_ref_fuse_broyden(C.c_int(B), C.c_int64(N), C.c_int(I), _p(xd_tgt), _p(voxel_J), C.c_int(D), C.c_int(H),
                              C.c_int(W), _p(tfs), _p(bone_ids), _p(offset), _p(scale), C.c_float(cvg),
                              C.c_float(dvg), _p(x), _p(Ji), _p(valid))
    return x, Ji, valid.astype(bool)


def filter(x, mask):
    x = _f32(x)
    mask = _u8(mask)
    B, N, I, _ = x.shape
    assert B == 1
    out = np.zeros((B, N, I), np.uint8)
    lib().ia_ref_filter(C.c_int64(N), C.c_int(I), _p(x), _p(mask), _p(out))
    return out.astype(bool)


# ----------------------------------------------------------------------------- fields
HASH_CFG = dict(n_levels=16, F=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.447269237440378)


def hashgrid_offsets(cfg=HASH_CFG):
    L = cfg["n_levels"]
    offs = np.zeros(L + 1, np.uint32)
    res = np.zeros(L, np.uint32)
    sc = np.zeros(L, np.float32)
    total = lib().ia_ref_hashgrid_offsets(C.c_int(L), C.c_int(cfg["log2_hashmap_size"]),
                                          C.c_int(cfg["base_resolution"]), C.c_float(cfg["per_level_scale"]),
                                          _p(offs), _p(res), _p(sc))
    return int(total), offs, res, sc


def hashgrid_fwd(x, params, cfg=HASH_CFG, with_jac=False):
    x, params = _f32(x), _f32(params)
    n = x.shape[0]
    LF = cfg["n_levels"] * cfg["F"]
    out = np.zeros((n, LF), np.float32)
    jac = np.zeros((n, LF, 3), np.float32) if with_jac else None
    lib().ia_ref_hashgrid_fwd(C.c_int64(n), _p(x), _p(params), C.c_int(cfg["n_levels"]), C.c_int(cfg["F"]),
                              C.c_int(cfg["log2_hashmap_size"]), C.c_int(cfg["base_resolution"]),
                              C.c_float(cfg["per_level_scale"]), _p(out), _p(jac))
    return (out, jac) if with_jac else out


def hashgrid_bwd_params(x, dL_dy, n_params, cfg=HASH_CFG):
    x, dL_dy = _f32(x), _f32(dL_dy)
    g = np.zeros(n_params, np.float32)
    lib().ia_ref_hashgrid_bwd_params(C.c_int64(x.shape[0]), _p(x), _p(dL_dy), C.c_int(cfg["n_levels"]),
                                     C.c_int(cfg["F"]), C.c_int(cfg["log2_hashmap_size"]),
                                     C.c_int(cfg["base_resolution"]), C.c_float(cfg["per_level_scale"]), _p(g))
    return g


def sh4(d01):
    d01 = _f32(d01)
    out = np.zeros((d01.shape[0], 16), np.float32)
    lib().ia_ref_sh4(C.c_int64(d01.shape[0]), _p(d01), _p(out))
    return out


def mlp_fwd(x, Ws, bs, hidden_act="relu", out_act="none"):
    x = _f32(x)
    Ws = [_f32(w) for w in Ws]
    bs = [_f32(b) for b in bs]
    nl = len(Ws)
    dims = _i32([Ws[0].shape[1]] + [w.shape[0] for w in Ws])
    Wp = (C.c_void_p * nl)(*[w.ctypes.data for w in Ws])
    bp = (C.c_void_p * nl)(*[b.ctypes.data for b in bs])
    out = np.zeros((x.shape[0], int(dims[-1])), np.float32)
    lib().ia_ref_mlp_fwd(C.c_int64(x.shape[0]), C.c_int(nl), _p(dims), Wp, bp,
                         C.c_int({"relu": 0, "softplus100": 1}[hidden_act]),
                         C.c_int({"none": 0, "sigmoid": 1}[out_act]), _p(x), _p(out))
    return out


def laplace_alpha(sdf, dists, beta):
    sdf, dists = _f32(sdf), _f32(dists)
    out = np.zeros_like(sdf)
    lib().ia_ref_laplace_alpha(C.c_int64(sdf.shape[0]), _p(sdf), _p(dists), C.c_float(beta), _p(out))
    return out


def sdf_field(x, center, scale, params, level_mask, W1, b1, W2, b2, with_grad=True, with_feat=True):
    x = _f32(x)
    n = x.shape[0]
    center, scale = _f32(np.asarray(center).reshape(3)), _f32(np.asarray(scale).reshape(3))
    params, level_mask = _f32(params), _f32(level_mask)
    W1, b1, W2, b2 = _f32(W1), _f32(b1), _f32(W2), _f32(b2)
    sdf = np.zeros(n, np.float32)
    grad = np.zeros((n, 3), np.float32) if with_grad else None
    feat = np.zeros((n, 13), np.float32) if with_feat else None
    lib().ia_ref_sdf_field(C.c_int64(n), _p(x), _p(center), _p(scale), _p(params), _p(level_mask), _p(W1), _p(b1),
                           _p(W2), _p(b2), _p(sdf), _p(grad), _p(feat))
    return sdf, grad, feat
