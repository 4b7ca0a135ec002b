#!/usr/bin/env python3
"""K9-consistent early filter of the Broyden search (ia_fuse_broyden_spec; snarf.hip) on the point distribution of the headline
step's secondary march (IA_POSE, default the bench's male-3-casual:0), in the spatial order the product path uses: per eps -- time,
fetches issued (counters of the kernel), points redone with the filter off, and how often the filter changes anything downstream:

  set_mismatch      points whose post-K9 candidate set (filter.cu:10-54) differs from the exact search's
  sdf_bits_differ   points whose min-over-candidates SDF differs in any bit
  sdf_abs_gt_1e-4   ... by more than 1e-4 (canonical metres) / 1e-3
  lost_root         points where the exact search keeps a candidate with no speculative candidate within 1 mm

Prints JSON (profiles/r04_spec_search_probe.json)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, render, nerfacc, fast_snarf, _lib as L

dev = "cuda:0"


def march_points(rs, rays, n_sec, seed=0):
    """sample points of n_sec secondary rays leaving the surface (the secondary march of the headline step)."""
    out = rs.forward(rays)
    hit = torch.nonzero(out["opacity"][:, 0] > 0.5)[:, 0]
    r = rs.deformer.transform_rays_w2s(rays.float())
    g = torch.Generator().manual_seed(seed)
    pick = hit[torch.randint(0, hit.shape[0], (n_sec,), generator=g).to(dev)]
    p = r[pick, :3] + r[pick, 3:6] * out["depth"][pick]
    d = torch.nn.functional.normalize(torch.randn((n_sec, 3), generator=g).to(dev), dim=-1)
    nrm = torch.nn.functional.normalize(out["comp_normal"][pick] @ rs.deformer.w2s[:3, :3].T, dim=-1)
    d = torch.where(((d * nrm).sum(-1) > 0)[:, None], d, -d).contiguous()
    iv, sm, _ = nerfacc.traverse_grids(p.contiguous(), d, rs.binaries, rs.aabbs, torch.zeros(n_sec, device=dev),
                                       torch.full((n_sec,), 1.5, device=dev), 1.5 / 63, 0.0, grid_bits=rs.grid_bits, max_extent=1.5)
    ts = iv.vals[iv.is_left]
    pts = render.ray_points(p.contiguous(), d, sm.ray_indices, ts)
    order = rs._spatial_order(pts)
    return pts[order.long()].contiguous()


def search(dfm, pts, eps, counters=None):
    P, I = pts.shape[0], dfm.init_bones.shape[0]
    x = torch.zeros((1, P, I, 3), device=dev)
    valid = torch.zeros((1, P, I), dtype=torch.bool, device=dev)
    vj = fast_snarf.ChannelLastVoxelJ(dfm.voxel_J_cl)
    if eps is None:
        fast_snarf.fuse_broyden(x, pts.reshape(1, P, 3), None, vj, dfm.tfs, dfm.init_bones, True, None, valid, dfm.offset_kernel,
                                dfm.scale_kernel, 1e-5, 1e-1)
    else:
        fast_snarf.fuse_broyden_spec(x, pts.reshape(1, P, 3), vj, dfm.tfs, dfm.init_bones, None, valid, dfm.offset_kernel,
                                     dfm.scale_kernel, 1e-5, 1e-1, eps, counters=counters, cell_tight=dfm.cell_tight)
    return x, valid


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def compare(dfm, geo, pts, eps, ref):
    """ref = (x, valid, keep, sdf) of the exact search."""
    x0, v0, k0, s0 = ref
    cnt = torch.zeros(5, dtype=torch.int64, device=dev)
    x1, v1 = search(dfm, pts, eps, counters=cnt)
    P, I = v0.shape[1:]
    # every item the speculative search completes is the exact search's item
    assert bool((v1 & ~v0).sum() == 0), "speculation produced a candidate the exact search does not have"
    same_x = torch.equal(torch.where(v1[..., None], x1, torch.zeros_like(x1)), torch.where(v1[..., None], x0, torch.zeros_like(x0)))
    k1 = fast_snarf.filter(x1, v1)
    set_mismatch = (k0 != k1).any(-1)[0]
    old = dfm.spec_eps, dfm.SPEC_MIN_POINTS
    try:
        dfm.spec_eps, dfm.SPEC_MIN_POINTS = eps, 0
        s1 = dfm.deform_sdf(pts, geo)
    finally:
        dfm.spec_eps, dfm.SPEC_MIN_POINTS = old
    dsdf = (s1 - s0).abs()
    # lost root: exact survivor with no speculative survivor within 1 mm
    xs0 = torch.where(k0[..., None], x0, torch.full_like(x0, 1e9))[0]
    xs1 = torch.where(k1[..., None], x1, torch.full_like(x1, -1e9))[0]
    lost = torch.zeros(P, dtype=torch.bool, device=dev)
    for i in range(I):
        dmin = (xs0[:, i:i + 1, :] - xs1).abs().amax(-1).amin(-1)
        lost |= k0[0, :, i] & (dmin > 1e-3)
    c = cnt.cpu().tolist()
    return dict(eps=eps, fetches=c[0], retired_items=c[1], completed_valid=c[2], redone_points=c[3], corner_loads=c[4],
                fetches_per_point=c[0] / P, completed_items_bit_identical=bool(same_x),
                set_mismatch=float(set_mismatch.float().mean()), sdf_bits_differ=float((s1 != s0).float().mean()),
                sdf_abs_gt_1e4=float((dsdf > 1e-4).float().mean()), sdf_abs_gt_1e3=float((dsdf > 1e-3).float().mean()),
                sdf_max_abs=float(dsdf.max()), lost_root=float(lost.float().mean()),
                survivors_per_point=float(k1.float().sum() / P))


def main():
    n_sec = int(os.environ.get("IA_NSEC", str(1 << 21)))
    pose = os.environ.get("IA_POSE", "male-3-casual:0")
    rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01, pose=(None if pose == "synthetic:0" else pose))
    pts = march_points(rs, rays, n_sec, seed=int(os.environ.get("IA_SEED", "0")))       # IA_SEED: another draw of the secondary rays
    if os.environ.get("IA_POINTS") == "box":          # another distribution: uniform in the frame's bounding box (most points far from the body)
        g = torch.Generator().manual_seed(int(os.environ.get("IA_SEED", "0")))
        lo, hi = rs.aabbs[0, :3], rs.aabbs[0, 3:]
        q = (torch.rand((pts.shape[0], 3), generator=g).to(dev) * (hi - lo) + lo).contiguous()
        pts = q[rs._spatial_order(q).long()].contiguous()
    elif os.environ.get("IA_POINTS") == "primary":    # the edges of the primary march (camera rays)
        r = rs.deformer.transform_rays_w2s(rays.float())
        ro, rd = r[:, :3].contiguous(), r[:, 3:6].contiguous()
        iv, sm, _ = nerfacc.traverse_grids(ro, rd, rs.binaries, rs.aabbs, torch.zeros(ro.shape[0], device=dev), torch.full((ro.shape[0],), 1e10, device=dev),
                                           rs.render_step_size, 0.0, grid_bits=rs.grid_bits)
        q = render.ray_points(ro, rd, iv.ray_indices, iv.vals)
        pts = q[rs._spatial_order(q).long()].contiguous()
    dfm, geo = rs.deformer, rs.geometry
    P = pts.shape[0]
    os.environ["IA_BROYDEN_SCHEDULE"] = "persistent"
    x0, v0 = search(dfm, pts, None)
    k0 = fast_snarf.filter(x0, v0)
    old = dfm.spec_eps
    dfm.spec_eps = 0.0
    s0 = dfm.deform_sdf(pts, geo)
    dfm.spec_eps = old
    cstat = torch.zeros(17, dtype=torch.int64, device=dev)
    _, _, D, H, W = dfm.lbs_voxel_final.shape
    L.check(L.lib().ia_broyden_stats(L.i32(1), L.i64(P), L.i32(13), L.ptr(pts), L.ptr(dfm.voxel_J_cl), L.i32(1), L.i32(D), L.i32(H), L.i32(W),
                                     L.ptr(dfm.tfs), L.ptr(dfm.init_bones), L.ptr(dfm.offset_kernel), L.ptr(dfm.scale_kernel),
                                     L.f32(1e-5), L.f32(1e-1), L.ptr(cstat), L.stream()), "ia_broyden_stats")
    cs = cstat.cpu().tolist()
    res = dict(pose=pose, seed=int(os.environ.get("IA_SEED", "0")), points=P, exact=dict(ms=timed(lambda: search(dfm, pts, None)), fetches=cs[0], corner_loads=cs[1], fetches_per_point=cs[0] / P,
                                    survivors_per_point=float(k0.float().sum() / P)), spec=[])
    for eps in [float(e) for e in os.environ.get("IA_EPS_LIST", "0,5e-4,1e-3,2e-3").split(",")]:
        r = compare(dfm, geo, pts, eps, (x0, v0, k0, s0))
        r["ms"] = timed(lambda: search(dfm, pts, eps))
        r["Gfetch_per_s"] = r["fetches"] / r["ms"] / 1e6
        r["speedup_vs_exact"] = res["exact"]["ms"] / r["ms"]
        res["spec"].append(r)
    res["exact"]["Gfetch_per_s"] = cs[0] / res["exact"]["ms"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
