#!/usr/bin/env python3
"""the SDF-only query path (XCD-partitioned hash gather -> level-major features -> ia_sdf_levels_fwd) on a big sorted batch: ms per
kernel.  Used with tools/pmc_probe.sh to read what the head waits for."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, _lib as L, fields
if os.environ.get("IA_PROBE_LOG2"):          # experiment: smaller tables (a prefix of the parameter vector) -- what does the L2 capacity cost?
    fields.HASH["log2_hashmap_size"] = int(os.environ["IA_PROBE_LOG2"])
dev = "cuda:0"
n = int(os.environ.get("IA_N", 100_000_000))
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01)
geo = rs.geometry
g = torch.Generator(device=dev).manual_seed(0)
x = (geo.center + (torch.rand((n, 3), device=dev, generator=g) - 0.5) * geo.scale * 0.5).contiguous()
x = x[rs._spatial_order(x).long()].contiguous()
for _ in range(2): y = geo.sdf_only(x)
lib = L.lib(); lib.start()
for _ in range(3): y = geo.sdf_only(x)
per = lib.report()
print(json.dumps({k: round(v[1] / v[0], 3) for k, v in per.items()} | dict(points=n)))
