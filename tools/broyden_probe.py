#!/usr/bin/env python3
"""Broyden search (ia_fuse_broyden) on the point distribution of the headline step's secondary march: time per schedule,
counted fetches (ia_broyden_stats), bytes through the vector-memory path.  Prints JSON."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, fields, pbr, render, nerfacc, fast_snarf, _lib as L

dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01)
n_sec = int(os.environ.get("IA_NSEC", str(1 << 21)))
# secondary rays of the headline step: surface points of hit primary rays + random outward directions
out = rs.forward(rays)
hit = torch.nonzero(out["opacity"][:, 0] > 0.5)[:, 0]
r = rs.deformer.transform_rays_w2s(rays.float())
g = torch.Generator().manual_seed(0)
pick = hit[torch.randint(0, hit.shape[0], (n_sec,), generator=g).to(dev)]
p = r[pick, :3] + r[pick, 3:6] * out["depth"][pick]
d = torch.nn.functional.normalize(torch.randn((n_sec, 3), generator=g).to(dev), dim=-1)
nrm = torch.nn.functional.normalize(out["comp_normal"][pick] @ rs.deformer.w2s[:3, :3].T, dim=-1)
d = torch.where(((d * nrm).sum(-1) > 0)[:, None], d, -d).contiguous()
step = 1.5 / 63
iv, sm, _ = nerfacc.traverse_grids(p.contiguous(), d, rs.binaries, rs.aabbs, torch.zeros(n_sec, device=dev),
                                   torch.full((n_sec,), 1.5, device=dev), step, 0.0, grid_bits=rs.grid_bits, max_extent=1.5)
ts = iv.vals[iv.is_left]
pts = render.ray_points(p.contiguous(), d, sm.ray_indices, ts)
P = pts.shape[0]
dfm = rs.deformer
I = 13
x = torch.empty((1, P, I, 3), device=dev); valid = torch.empty((1, P, I), dtype=torch.bool, device=dev)
vj = fast_snarf.ChannelLastVoxelJ(dfm.voxel_J_cl)
def run():
    fast_snarf.fuse_broyden(x, pts.reshape(1, P, 3), None, vj, dfm.tfs, dfm.init_bones, True, None, valid, dfm.offset_kernel,
                            dfm.scale_kernel, 1e-5, 1e-1)
res = dict(points=P, samples_per_ray=P / n_sec)
_, _, D, H, W = dfm.lbs_voxel_final.shape
ref_out = None
for sched, v2 in (("persistent", "0"), ("persistent", "1"), ("simple", "1")):
    os.environ["IA_BROYDEN_SCHEDULE"], os.environ["IA_BROYDEN_V2"] = sched, v2
    x.fill_(0); valid.fill_(False)
    run(); torch.cuda.synchronize()
    cur = (valid.clone(), torch.where(valid[..., None], x, torch.zeros_like(x)))
    if ref_out is None:
        ref_out = cur
    else:
        assert torch.equal(cur[0], ref_out[0]) and torch.equal(cur[1], ref_out[1]), "schedules differ"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    res[sched + ("_v2" if (v2 == "1" and sched == "persistent") else "") + "_ms"] = e0.elapsed_time(e1) / 3
os.environ["IA_BROYDEN_V2"] = os.environ.get("IA_PROBE_V2", "1")
def stats_for(points):
    c = torch.zeros(17, dtype=torch.int64, device=dev)
    L.check(L.lib().ia_broyden_stats(L.i32(1), L.i64(points.shape[0]), L.i32(I), L.ptr(points), L.ptr(dfm.voxel_J_cl), L.i32(1), L.i32(D), L.i32(H),
                                     L.i32(W), L.ptr(dfm.tfs), L.ptr(dfm.init_bones), L.ptr(dfm.offset_kernel), L.ptr(dfm.scale_kernel),
                                     L.f32(1e-5), L.f32(1e-1), L.ptr(c), L.stream()), "ia_broyden_stats")
    return c.cpu().tolist()
if os.environ.get("IA_ORDER_EXPERIMENT", "1") == "1":
    # how much does the ORDER / locality of the items matter?  same multiset of points: ray order (above), random order,
    # spatially sorted (Morton code of the 1 cm cell), and the degenerate all-items-identical case (pure L1-hit rate)
    os.environ["IA_BROYDEN_SCHEDULE"] = "persistent"
    base_pts = pts
    q = ((base_pts - base_pts.min(0)[0]) / 0.01).long().clamp(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; return (v | (v << 2)) & 0x09249249
    morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    variants = dict(random=base_pts[torch.randperm(P, device=dev)].contiguous(), morton=base_pts[torch.argsort(morton)].contiguous(),
                    identical=base_pts[P // 2:P // 2 + 1].expand(P, 3).contiguous())
    for name, v in variants.items():
        pts = v
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        c = stats_for(v)
        res["order_" + name] = dict(ms=ms, fetches=c[0], ns_per_fetch_lane=ms * 1e6 / max(c[0], 1), Gfetch_per_s=c[0] / ms / 1e6)
    pts = base_pts
cnt = torch.zeros(17, dtype=torch.int64, device=dev)
L.check(L.lib().ia_broyden_stats(L.i32(1), L.i64(P), L.i32(I), L.ptr(pts), L.ptr(dfm.voxel_J_cl), L.i32(1), L.i32(D), L.i32(H), L.i32(W),
                                 L.ptr(dfm.tfs), L.ptr(dfm.init_bones), L.ptr(dfm.offset_kernel), L.ptr(dfm.scale_kernel),
                                 L.f32(1e-5), L.f32(1e-1), L.ptr(cnt), L.stream()), "ia_broyden_stats")
c = cnt.cpu().tolist()
best = min(res["persistent_ms"], res["simple_ms"])
res.update(fetches=c[0], corner_loads=c[1], converged=c[2], diverged=c[3], exhausted=c[4], fetches_per_item=c[0] / (P * I),
           corners_per_fetch=c[1] / max(c[0], 1), exit_after_k_fetches={k: c[5 + k] for k in range(12)},
           l1_TBps_requested=c[1] * 48 / (best * 1e-3) / 1e12, ns_per_point=best * 1e6 / P,
           valid_frac=float(valid.float().mean()))
print(json.dumps(res))
