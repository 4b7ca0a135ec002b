#!/usr/bin/env python3
"""The runtime canary of the early-filter search (SNARFDeformer.spec_canary) on all eight committed frames of the reference's pose files:
one relit 270 x 270 frame per pose (render_mode = light, spp 256, GI on: every search batch of the path -- primary it0 / it1, shading,
secondary march, secondary shading -- goes through _candidates), every 16th point of every batch searched again to the end + K9.
Prints per pose: points checked, candidate rows that differ (must be 0), overflow points (count compared only)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, fields, pbr

dev = "cuda:0"
hw, spp, every = int(os.environ.get("IA_HW", "270")), 256, int(os.environ.get("IA_CANARY", "16"))
mat = fields.VolumeMaterial(seed=2).to(dev)
v, u = np.meshgrid((np.arange(64) + 0.5) / 64, (np.arange(128) + 0.5) / 128, indexing="ij")
img = np.where((v < 0.5)[..., None], np.stack([0.3 + 0.4 * (1 - v), 0.4 + 0.4 * (1 - v), 0.6 + 0.4 * (1 - v)], -1), 0.08)
img = img + (50.0 * np.exp(-(((u - 0.3) * 2) ** 2 + ((v - 0.25) * 2) ** 2) / (2 * 0.05 ** 2)))[..., None]
env = pbr.EnvironmentLightTensor(torch.from_numpy(img.astype(np.float32)).to(dev)); env.update_pdf()
g = torch.Generator().manual_seed(0)
light_u = torch.rand((spp, 3), generator=g).to(dev)
rows = []
for pose in ("male-3-casual:0", "male-3-casual:40", "male-3-casual:80", "male-3-casual:113", "aist:0", "aist:100", "aist:200", "aist:319"):
    rs, rays, _ = S.build_frame(dev, hw, hw, pose=pose, beta=0.01, num_samples_per_ray=128)
    rs.deformer.spec_canary = every
    rs.deformer.canary_totals(reset=True)
    shuffle_u = torch.rand((rays.shape[0], spp), generator=g).to(dev)
    out = rs.relight(rays, mat, env, spp, light_u, shuffle_u, background_color=torch.ones(3, device=dev), global_illumination=True)
    n, bad, ovf = rs.deformer.canary_totals()
    rows.append(dict(pose=pose, every=every, points_checked=n, candidate_rows_differ=bad, overflow_points_count_only=ovf,
                     secondary_rays=int(out["stats"]["n_secondary"])))
    print(json.dumps(rows[-1]), flush=True)
    del rs, out
    torch.cuda.empty_cache()
print(json.dumps(dict(total_points=sum(r["points_checked"] for r in rows), total_differ=sum(r["candidate_rows_differ"] for r in rows))))
