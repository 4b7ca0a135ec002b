#!/usr/bin/env python3
"""search time against batch size (50 k .. 2.3 M march points): exact item-per-lane kernel (ia_fuse_broyden) vs the per-point
speculative kernel (ia_fuse_broyden_spec).  The reference trains on 4096 rays per GPU: ~1 M points per search call."""
import os, sys, json, torch
sys.path.insert(0, "/root/repo")
from intrinsicavatar_amd import build; build.build()
from tools import spec_search_probe as SP
from intrinsicavatar_amd import synthetic as S
dev="cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01)
pts = SP.march_points(rs, rays, 1 << 18)
res={}
for n in (50_000, 230_000, 1_000_000, pts.shape[0]):
    sub = pts[:n].contiguous()
    res[n]=dict(exact_ms=round(SP.timed(lambda: SP.search(rs.deformer, sub, None), 5),3), spec_ms=round(SP.timed(lambda: SP.search(rs.deformer, sub, 1e-3), 5),3))
print(json.dumps(res))
