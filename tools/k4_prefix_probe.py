#!/usr/bin/env python3
"""How many of the secondary march's samples does K4 (ray_resampling_sdf_fine, cdf.cu:536-636) actually READ?
K4 walks a ray's samples to its FIRST zero crossing (sdf[i] >= 0 and sdf[i+1] < 0), then inverts the CDF of the alphas from that
interval on until its five targets u = 0.1 ... 0.9 are placed -- it never looks at a sample behind that point, and nothing else
consumes the coarse SDF of the secondary march (models/intrinsic_avatar.py:396-545).  Rays without a crossing read every sample.
The probe captures (packed_info, alphas, sdf) of every K4 call of one relit frame and replays the walk: samples per ray, the prefix
K4 needs, what is dead.     python tools/k4_prefix_probe.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, fields, pbr, lib_nerfacc

dev = "cuda:0"
hw, spp = int(os.environ.get("IA_HW", "540")), int(os.environ.get("IA_SPP", "128"))
rs, rays, _ = S.build_frame(dev, hw, hw, pose=os.environ.get("IA_POSE", "male-3-casual:0"), beta=0.01, num_samples_per_ray=128)
mat = fields.VolumeMaterial(seed=2).to(dev)
v, u = np.meshgrid((np.arange(64) + 0.5) / 64, (np.arange(128) + 0.5) / 128, indexing="ij")
img = np.where((v < 0.5)[..., None], np.stack([0.3 + 0.4 * (1 - v), 0.4 + 0.4 * (1 - v), 0.6 + 0.4 * (1 - v)], -1), 0.08)
env = pbr.EnvironmentLightTensor(torch.from_numpy(img.astype(np.float32)).to(dev)); env.update_pdf()
g = torch.Generator().manual_seed(0)
light_u = torch.rand((spp, 3), generator=g).to(dev)
shuffle_u = torch.rand((rays.shape[0], spp), generator=g).to(dev)
captured = []
orig = lib_nerfacc.ray_resampling_sdf_fine
def hook(pinfo, ts, te, alphas, sdfs, n):
    captured.append((pinfo.clone(), alphas.clone(), sdfs.clone()))
    return orig(pinfo, ts, te, alphas, sdfs, n)
lib_nerfacc.ray_resampling_sdf_fine = hook
os.environ["IA_SECONDARY_STREAMS"] = "1"
type(rs).SECONDARY_STREAMS = 1
out = rs.relight(rays, mat, env, spp, light_u, shuffle_u, background_color=torch.ones(3, device=dev), global_illumination=True)
tot = dict(rays=0, rays_with_samples=0, samples=0, needed=0, rays_with_crossing=0)
hist_steps = torch.zeros(65, dtype=torch.long)
hist_need = torch.zeros(65, dtype=torch.long)
for pinfo, alphas, sdf in captured:
    base, steps = pinfo[:, 0].long(), pinfo[:, 1].long()
    n = base.shape[0]
    smax = int(steps.max())
    k = torch.arange(smax, device=dev)[None, :]
    ok = k < steps[:, None]
    idx = (base[:, None] + k).clamp(max=sdf.shape[0] - 1)
    sd = torch.where(ok, sdf[idx], torch.full((), 1e9, device=dev))
    al = torch.where(ok, alphas[idx], torch.zeros((), device=dev))
    cross = (sd[:, :-1] >= 0) & (sd[:, 1:] < 0) & ok[:, 1:]
    has = cross.any(1)
    c = torch.where(has, cross.float().argmax(1), torch.zeros_like(steps))
    # CDF walk from interval c: cdf after interval m = 1 - prod_{c..m} (1 - alpha); the last target is u = 0.9
    one_m = torch.where(k >= c[:, None], 1 - al, torch.ones_like(al))
    cdf = 1 - torch.cumprod(one_m, 1)
    reach = (cdf > 0.9) & ok & (k >= c[:, None])
    last = torch.where(reach.any(1), reach.float().argmax(1), steps - 1)
    need = torch.where(has, torch.maximum(last + 1, c + 2), steps).clamp(max=steps)
    tot["rays"] += n; tot["rays_with_samples"] += int((steps > 0).sum()); tot["samples"] += int(steps.sum()); tot["needed"] += int(need.sum())
    tot["rays_with_crossing"] += int(has.sum())
    hist_steps += torch.bincount(steps.clamp(max=64).cpu(), minlength=65)
    hist_need += torch.bincount(need.clamp(max=64).cpu(), minlength=65)
cum = lambda h: [round(float(x), 4) for x in (torch.cumsum(h, 0).float() / h.sum())[[1, 2, 4, 8, 16, 32, 64]]]
print(json.dumps(dict(pose=os.environ.get("IA_POSE", "male-3-casual:0"), spp=spp, **tot, dead_fraction=round(1 - tot["needed"] / max(tot["samples"], 1), 4),
                      samples_per_ray=round(tot["samples"] / max(tot["rays"], 1), 2), needed_per_ray=round(tot["needed"] / max(tot["rays"], 1), 2),
                      cdf_of_samples_per_ray_at_1_2_4_8_16_32_64=cum(hist_steps), cdf_of_needed_per_ray=cum(hist_need))))
