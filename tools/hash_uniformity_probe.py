#!/usr/bin/env python3
"""How coherent is a wavefront of the hash gather?  The candidates of ~16 M sorted march points in the product's order (point-major:
close to the split order for this purpose), in groups of 64 consecutive list entries (= the lanes of one wave of hash_fwd_xcd_kernel):
per level, the share of waves whose 64 points fall into ONE cell / <= 2 / <= 4 cells, and the mean number of distinct cells.
python tools/hash_uniformity_probe.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, fields
from tools import spec_search_probe as SP
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose="male-3-casual:0", beta=0.01, num_samples_per_ray=128)
pts = SP.march_points(rs, rays, 1 << 21)
geo, dfm = rs.geometry, rs.deformer
r = dfm._candidates(pts, with_src=False, normalize=(geo.center, geo.scale), split=True)
cand_x, Q = r[0], r[4]
H = fields.HASH
L_, base, pls = H["n_levels"], H["base_resolution"], H["per_level_scale"]
n = (Q // 64) * 64
x = cand_x[:n]
out = dict(candidates=int(Q), levels=[])
import math
for l in range(L_):
    sc = math.pow(2.0, l * math.log2(pls)) * base - 1.0          # tiny-cuda-nn: scale = 2^(l log2 s) base - 1, res = ceil(scale) + 1
    g = torch.floor(x * sc + 0.5).to(torch.int64)
    key = (g[:, 0] + 4096 * (g[:, 1] + 4096 * g[:, 2])).reshape(-1, 64)
    srt = torch.sort(key, dim=1).values
    distinct = 1 + (srt[:, 1:] != srt[:, :-1]).sum(1)
    out["levels"].append(dict(level=l, res=int(math.ceil(sc)) + 1, one=round(float((distinct == 1).float().mean()), 4),
                              le2=round(float((distinct <= 2).float().mean()), 4), le4=round(float((distinct <= 4).float().mean()), 4),
                              mean_distinct=round(float(distinct.float().mean()), 2)))
print(json.dumps(out))
