#!/usr/bin/env python3
"""Per-kernel timing of the hot-path operators at BASELINE config-2 sizes (540x540 frame).
Used with rocprofv3 to produce the per-round profiles under profiles/."""
import json
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build, synthetic as S   # noqa: E402

build.build()
from intrinsicavatar_amd import nerfacc, lib_nerfacc, fast_snarf, _lib as L   # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    hw = int(os.environ.get("IA_HW", "540"))
    res = {}
    sc = S.make_scene(hw, hw, pose_seed=0)
    rays = torch.from_numpy(sc["rays"]).to(DEV)
    n = rays.shape[0]
    ro, rd = rays[:, :3].contiguous(), rays[:, 3:6].contiguous()
    binaries = torch.from_numpy(sc["binaries"]).to(DEV)[None]
    aabb = torch.from_numpy(sc["aabb"]).to(DEV)[None]
    near = torch.zeros(n, device=DEV)
    far = torch.full((n,), 1e10, device=DEV)
    step = 4.3301 / 128
    bits = nerfacc.pack_occupancy_bits(binaries[0])
    lib, st = L.lib(), L.stream()
    scratch = torch.empty(int(lib.ia_traverse_scratch_bytes(L.i64(n))), dtype=torch.uint8, device=DEV)
    pcnt = torch.empty(n, dtype=torch.int64, device=DEV)
    pstart = torch.empty(n, dtype=torch.int64, device=DEV)
    tot = torch.zeros(1, dtype=torch.int64, device=DEV)
    tmp = L.scan_tmp(n, DEV)
    aabb0 = aabb[0].contiguous()
    args = (L.i64(n), L.ptr(ro), L.ptr(rd), L.ptr(bits), 64, 64, 64, L.ptr(aabb0), L.ptr(near), L.ptr(far), L.f32(step), L.f32(0.0))

    def count():
        L.check(lib.ia_traverse_grids_count(*args, L.ptr(scratch), L.ptr(pcnt), st))

    def scan():
        lib.ia_exclusive_scan_i64(L.ptr(pcnt), L.ptr(pstart), L.ptr(tot), L.i64(n), L.ptr(tmp), st)

    count(); scan()
    t_ = int(tot.item())
    E, S_ = t_ & 0xFFFFFFFF, t_ >> 32
    iv_vals = torch.empty(E, device=DEV)
    fl = torch.zeros((2, E), dtype=torch.bool, device=DEV)
    iv_ray = torch.empty(E, dtype=torch.int64, device=DEV)
    sm_vals = torch.empty(S_, device=DEV)
    sm_ray = torch.empty(S_, dtype=torch.int64, device=DEV)
    term = torch.empty(n, device=DEV)
    pinfo = torch.empty((2, n, 2), dtype=torch.int64, device=DEV)

    def fill():
        L.check(lib.ia_traverse_grids_fill(*args, L.ptr(scratch), L.ptr(pcnt), L.ptr(pstart), L.ptr(pinfo[0]), L.ptr(pinfo[1]),
                                           L.ptr(iv_vals), L.ptr(fl[0]), L.ptr(fl[1]), L.ptr(iv_ray), L.ptr(sm_vals), L.ptr(sm_ray),
                                           L.ptr(term), st))
    t_count, t_scan, t_fill = timeit(count), timeit(scan), timeit(fill)
    alg_bytes = 48 * n + 16 * S_ + 14 * E
    res["traverse"] = dict(n_rays=n, E=E, S=S_, count_us=t_count, scan_us=t_scan, fill_us=t_fill,
                           alg_bytes=alg_bytes, gbps_fill=alg_bytes / t_fill / 1e3,
                           gbps_total=alg_bytes / (t_count + t_scan + t_fill) / 1e3)

    # ---- single-launch traversal, primary rays and a secondary-march batch (2M rays leaving occupied cells, [0, 1.5])
    def fused_case(tag, ro_, rd_, near_, far_, step_, extent):
        nonlocal bits
        m = ro_.shape[0]
        a_ = (L.i64(m), L.ptr(ro_), L.ptr(rd_), L.ptr(bits), 64, 64, 64, L.ptr(aabb0), L.ptr(near_), L.ptr(far_), L.f32(step_), L.f32(0.0))
        smax = int(np.ceil(extent / step_)) + 2
        cs, ce = m * smax, m * smax + 8 * m
        fs = torch.empty(int(lib.ia_traverse_fused_scratch_bytes(L.i64(m))), dtype=torch.uint8, device=DEV)
        totals = torch.empty(3, dtype=torch.int64, device=DEV)
        b_iv, b_fl, b_ir = torch.empty(ce, device=DEV), torch.empty((2, ce), dtype=torch.bool, device=DEV), torch.empty(ce, dtype=torch.int64, device=DEV)
        b_sv, b_sr = torch.empty(cs, device=DEV), torch.empty(cs, dtype=torch.int64, device=DEV)
        b_t, b_pi = torch.empty(m, device=DEV), torch.empty((2, m, 2), dtype=torch.int64, device=DEV)

        def fused():
            L.check(lib.ia_traverse_grids_fused(*a_, L.ptr(fs), L.i64(ce), L.i64(cs), L.ptr(totals), L.ptr(b_pi[0]), L.ptr(b_pi[1]),
                                                L.ptr(b_iv), L.ptr(b_fl[0]), L.ptr(b_fl[1]), L.ptr(b_ir), L.ptr(b_sv), L.ptr(b_sr),
                                                L.ptr(None if "no_planes" in tag else b_t), L.ptr(None), L.ptr(None),
                                                L.i32(1 if "secondary" in tag else 0), st))
        us = timeit(fused)
        E_, S__, ovf = totals.tolist()
        ab = 48 * m + 16 * S__ + 14 * E_
        res[tag] = dict(n_rays=m, E=E_, S=S__, overflow=ovf, us=us, alg_bytes=ab, gbps=ab / us / 1e3, frac_of_8TBps=ab / us / 1e3 / 8000)

    fused_case("traverse_fused_primary", ro, rd, near, far, step, 4.3301)
    fused_case("traverse_fused_primary_no_planes", ro, rd, near, far, step, 4.3301)       # what render_step asks for: walk ends at the occupied box
    M = int(os.environ.get("IA_SECONDARY", "2097152"))
    gsec = torch.Generator().manual_seed(3)
    occ = torch.nonzero(binaries[0])                     # [K,3] occupied cells
    pick = occ[torch.randint(0, occ.shape[0], (M,), generator=gsec).to(DEV)]
    # consecutive rays leave neighbouring surface points (as the shading points of one pixel row do): sort by cell
    pick = pick[torch.argsort(pick[:, 0] * 4096 + pick[:, 1] * 64 + pick[:, 2])]
    cellsz = (aabb0[3:] - aabb0[:3]) / 64
    so = (aabb0[:3] + (pick.float() + torch.rand((M, 3), generator=gsec).to(DEV)) * cellsz).contiguous()
    sd = torch.nn.functional.normalize(torch.randn((M, 3), generator=gsec), dim=-1).to(DEV).contiguous()
    fused_case("traverse_fused_secondary", so, sd, torch.zeros(M, device=DEV), torch.full((M,), 1.5, device=DEV), 1.5 / 63, 1.5)
    fused_case("traverse_fused_secondary_no_planes", so, sd, torch.zeros(M, device=DEV), torch.full((M,), 1.5, device=DEV), 1.5 / 63, 1.5)
    # ---- what bounds the secondary march: the same 2 M rays (a) through an EMPTY grid -- set-up + the DDA over every
    # cell of the crossing, nothing emitted --, (b) started far outside the box pointing away -- set-up only --, (c) through a FULL grid
    # -- every step emits a sample.  (a) is the floor of any kernel that walks the cell sequence per ray; see DESIGN 4.2.
    bits_save2 = bits
    bits = nerfacc.pack_occupancy_bits(torch.zeros_like(binaries)[0])
    fused_case("traverse_fused_secondary_empty_grid", so, sd, torch.zeros(M, device=DEV), torch.full((M,), 1.5, device=DEV), 1.5 / 63, 1.5)
    far_o = (so + 100.0).contiguous()
    fused_case("traverse_fused_secondary_missing_the_box", far_o, sd.abs().contiguous(), torch.zeros(M, device=DEV), torch.full((M,), 1.5, device=DEV), 1.5 / 63, 1.5)
    bits = nerfacc.pack_occupancy_bits(torch.ones_like(binaries)[0])
    fused_case("traverse_fused_secondary_full_grid", so, sd, torch.zeros(M, device=DEV), torch.full((M,), 1.5, device=DEV), 1.5 / 63, 1.5)
    bits = bits_save2
    zn, fr = torch.zeros(M, device=DEV), torch.full((M,), 1.5, device=DEV)
    for meth in ("two_pass", "fused"):      # whole operator incl. allocation and the size sync
        res[f"traverse_op_secondary_{meth}"] = dict(us=timeit(lambda: nerfacc.traverse_grids(
            so, sd, binaries, aabb, zn, fr, 1.5 / 63, 0.0, grid_bits=bits, max_extent=1.5, method=meth), iters=5))
        res[f"traverse_op_primary_{meth}"] = dict(us=timeit(lambda: nerfacc.traverse_grids(
            ro, rd, binaries, aabb, near, far, step, 0.0, grid_bits=bits, method=meth), iters=5))
    full = torch.ones_like(binaries)
    bits_full = nerfacc.pack_occupancy_bits(full[0])
    bits_save, bits = bits, bits_full
    fused_case("traverse_fused_primary_dense_grid", ro, rd, near, far, step, 4.3301)
    bits = bits_save

    # ---- K2 merge on the traversal's edge list + T2
    iv_pi = pinfo[0].int().contiguous()
    w_e = torch.rand(E, device=DEV) * 0.1
    t_k2 = timeit(lambda: lib_nerfacc.ray_resampling_merge(iv_pi, iv_vals, fl[0], fl[1], w_e, 16))
    res["k2_merge"] = dict(E=E, us=t_k2)
    al = torch.rand(E, device=DEV) * 0.3
    wout, tout = torch.empty_like(al), torch.empty_like(al)
    t_t2 = timeit(lambda: lib.ia_render_weight_from_alpha(L.i64(n), L.ptr(iv_pi), L.ptr(al), L.ptr(wout), L.ptr(tout), st))
    res["t2_weights"] = dict(E=E, us=t_t2, gbps=E * 12 / t_t2 / 1e3)

    # ---- Broyden on the edge positions
    w, offk, sck, bbox = S.skinning_weight_grid(D=32, H=128, W=128, smooth_iters=2)
    vw = torch.from_numpy(w).to(DEV)
    tfs = torch.from_numpy(sc["rig"]["tfs"]).to(DEV)
    off, scl = torch.from_numpy(offk).to(DEV), torch.from_numpy(sck).to(DEV)
    vd = torch.zeros(1, 3, 32, 128, 128, device=DEV)
    vJ = torch.zeros(1, 12, 32, 128, 128, device=DEV)
    vJcl = torch.zeros(1, 32, 128, 128, 12, device=DEV)
    t_pre = timeit(lambda: fast_snarf.precompute(vw, tfs, vd, vJ, off, scl, voxel_J_cl=vJcl))
    res["precompute"] = dict(us=t_pre, gbps=(50.3e6 + 25.2e6 * 2 + 6.3e6) / t_pre / 1e3)
    pts = (ro[iv_ray] + rd[iv_ray] * iv_vals[:, None])[None].contiguous()
    P = pts.shape[1]
    x = torch.zeros(1, P, 13, 3, device=DEV)
    Ji = torch.zeros(1, P, 13, 3, 3, device=DEV)
    valid = torch.zeros(1, P, 13, dtype=torch.bool, device=DEV)
    bones = torch.from_numpy(S.INIT_BONES).to(DEV)
    for name, grid in (("broyden_ncdhw", vJ), ("broyden_ndhwc", fast_snarf.ChannelLastVoxelJ(vJcl))):
        t_b = timeit(lambda: fast_snarf.fuse_broyden(x, pts, vd, grid, tfs, bones, True, Ji, valid, off, scl, 1e-5, 1e-1), iters=5)
        res[name] = dict(P=P, us=t_b, Mpts_per_s=P / t_b, valid_frac=float(valid.float().mean()))
    t_f = timeit(lambda: fast_snarf.filter(x, valid), iters=5)
    res["filter"] = dict(P=P, us=t_f)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
