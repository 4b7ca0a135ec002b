#!/usr/bin/env python3
"""BASELINE config 3 / 5 shape: relight one 540x540 frame, render_mode=light, spp light samples per pixel,
secondary rays on.  Prints JSON (primary rays/s, secondary rays/s, per-entry-point breakdown)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, fields, pbr, _lib as L

hw = int(os.environ.get("IA_HW", "540")); spp = int(os.environ.get("IA_SPP", "256")); gi = os.environ.get("IA_GI", "0") == "1"
chunk = int(os.environ.get("IA_RAY_CHUNK", "65536"))
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, hw, hw, pose_seed=0, beta=0.01)
mat = fields.VolumeMaterial(seed=2).to(dev)
H, W = 1024, 2048
v, u = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
img = np.where((v < 0.5)[..., None], np.stack([0.3 + 0.4 * (1 - v), 0.4 + 0.4 * (1 - v), 0.6 + 0.4 * (1 - v)], -1), 0.08)
img = img + (5e4 * np.exp(-(((u - 0.3) * 2) ** 2 + ((v - 0.25) * 2) ** 2) / (2 * 0.01 ** 2)))[..., None]
env = pbr.EnvironmentLightTensor(torch.from_numpy(img.astype(np.float32)).to(dev)); env.update_pdf()
g = torch.Generator().manual_seed(0)
light_u = torch.rand((spp, 3), generator=g).to(dev)
n = rays.shape[0]
def frame():
    tot = dict(n_secondary=0, n_fg=0)
    for c0 in range(0, n, chunk):                 # ray chunks as the reference does at eval (ray_chunk), but 16x larger
        r = rays[c0:c0 + chunk]
        su = torch.rand((r.shape[0], spp), device=dev)
        o = rs.relight(r, mat, env, spp, light_u, su, global_illumination=gi)
        tot["n_secondary"] += o["stats"]["n_secondary"]; tot["n_fg"] += o["stats"]["n_fg"]
    return tot
frame(); torch.cuda.synchronize()
lib = L.lib(); lib.start(); t0 = time.perf_counter()
tot = frame(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
pc = lib.report()
print(json.dumps(dict(hw=hw, spp=spp, gi=gi, ray_chunk=chunk, s_per_frame=round(dt, 3), primary_rays_per_s=round(n / dt, 1),
                      secondary_rays=tot["n_secondary"], secondary_rays_per_s=round(tot["n_secondary"] / dt, 1), fg_points=tot["n_fg"],
                      breakdown_ms={k: round(v[1], 1) for k, v in sorted(pc.items(), key=lambda kv: -kv[1][1])[:8]},
                      kernel_ms=round(sum(v[1] for v in pc.values()), 1))))
