#!/usr/bin/env python3
"""A/B of the SDF value head variants (IA_SDF_HEAD): bitwise comparison on ragged sizes + timing on a big sorted batch."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, _lib as L
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 64, 64, pose_seed=0, beta=0.01)
geo = rs.geometry
g = torch.Generator(device=dev).manual_seed(0)
variants = os.environ.get("IA_VARIANTS", "pipe12,pipe2w12,pipe2w8,pipe2").split(",")
res = {}
for n in (1, 31, 32, 33, 63, 64, 65, 1000, 12345, (1 << 20) + 7, 5_000_011):
    x = (geo.center + (torch.rand((n, 3), device=dev, generator=g) - 0.5) * geo.scale * 0.9).contiguous()
    outs = []
    for v in variants:
        os.environ["IA_SDF_HEAD"] = v
        outs.append(geo.sdf_only(x).clone())
    ref = geo.forward(x, with_grad=False, with_feature=False)
    res[n] = dict(bit_equal=[bool(torch.equal(outs[0], o)) for o in outs[1:]], max_abs_vs_forward=float((outs[-1] - ref).abs().max()))
n = int(os.environ.get("IA_N", 50_000_000))
x = (geo.center + (torch.rand((n, 3), device=dev, generator=g) - 0.5) * geo.scale * 0.5).contiguous()
x = x[rs._spatial_order(x).long()].contiguous()
ms = {}
for v in variants:
    os.environ["IA_SDF_HEAD"] = v
    for _ in range(2): geo.sdf_only(x)
    lib = L.lib(); lib.start()
    for _ in range(3): geo.sdf_only(x)
    per = lib.report()
    ms[v] = round(per["ia_sdf_levels_fwd"][1] / per["ia_sdf_levels_fwd"][0], 3)
print(json.dumps(dict(parity=res, ms_per_call=ms, points=n)))
