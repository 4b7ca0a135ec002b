#!/usr/bin/env python3
"""distinct voxels per wave in the first two fetches of the deformer search (ia_broyden_voxel_stats) on the headline step's
secondary-march points, in ray order and in spatial (Morton) order, items point-major and init-major.  Prints JSON."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, render, nerfacc, _lib as L
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01)
n_sec = 1 << 21
out = rs.forward(rays)
hit = torch.nonzero(out["opacity"][:, 0] > 0.5)[:, 0]
r = rs.deformer.transform_rays_w2s(rays.float())
g = torch.Generator().manual_seed(0)
pick = hit[torch.randint(0, hit.shape[0], (n_sec,), generator=g).to(dev)]
p = r[pick, :3] + r[pick, 3:6] * out["depth"][pick]
d = torch.nn.functional.normalize(torch.randn((n_sec, 3), generator=g).to(dev), dim=-1)
nrm = torch.nn.functional.normalize(out["comp_normal"][pick] @ rs.deformer.w2s[:3, :3].T, dim=-1)
d = torch.where(((d * nrm).sum(-1) > 0)[:, None], d, -d).contiguous()
iv, sm, _ = nerfacc.traverse_grids(p.contiguous(), d, rs.binaries, rs.aabbs, torch.zeros(n_sec, device=dev), torch.full((n_sec,), 1.5, device=dev),
                                   1.5 / 63, 0.0, grid_bits=rs.grid_bits, max_extent=1.5)
pts = render.ray_points(p.contiguous(), d, sm.ray_indices, sm.t_starts)
order = rs._spatial_order(pts).long()
dfm = rs.deformer
_, _, D, H, W = dfm.lbs_voxel_final.shape
res = dict(points=pts.shape[0])
for oname, P in (("ray_order", pts), ("spatial_order", pts[order].contiguous())):
    for mode in (0, 1):
        c = torch.zeros(16, dtype=torch.int64, device=dev)
        L.check(L.lib().ia_broyden_voxel_stats(L.i64(P.shape[0]), L.i32(13), L.i32(mode), L.ptr(P), L.ptr(dfm.voxel_J_cl), L.i32(D), L.i32(H), L.i32(W),
                                               L.ptr(dfm.tfs), L.ptr(dfm.init_bones), L.ptr(dfm.offset_kernel), L.ptr(dfm.scale_kernel), L.ptr(c),
                                               L.stream()), "stats")
        c = c.cpu().tolist()
        for f in (0, 1):
            w = max(c[f * 8], 1)
            res[f"{oname}/{'init_major' if mode else 'point_major'}/fetch{f + 1}"] = dict(
                mean_distinct_voxels=round(c[f * 8 + 1] / w, 2),
                share_by_distinct_count={k: round(c[f * 8 + 2 + i] / w, 3) for i, k in enumerate(("1", "2", "3-4", "5-8", "9-16", "17+"))})
print(json.dumps(res, indent=1))
