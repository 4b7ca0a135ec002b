#!/usr/bin/env python3
"""BASELINE config 4 shape: training step WITH the PBR branch (uniform_light, spp 512) on a batch of 4096*G rays of one
540x540 frame (the reference trains on 4096 rays per GPU).  Prints JSON: ms/step, rays/s, secondary rays, breakdown."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, fields, pbr, _lib as L

dev = "cuda:0"
n_batch = int(os.environ.get("IA_BATCH", "4096"))
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01)
mat = fields.VolumeMaterial(seed=2).to(dev)
yy, xx = np.meshgrid(np.linspace(0, np.pi, 256), np.linspace(-np.pi, np.pi, 512), indexing="ij")
sky = (0.6 + 0.35 * np.cos(yy)[..., None] * np.array([1.0, 0.8, 0.6]) + 0.05 * np.sin(2 * xx)[..., None]).astype(np.float32)
env = pbr.EnvironmentLightTensor(torch.from_numpy(sky).to(dev)); env.update_pdf()
env_base = env.base.detach().clone().requires_grad_(True)
g = torch.Generator().manual_seed(0)
hit = torch.nonzero(rs.forward(rays)["opacity"][:, 0] > 0.5)[:, 0]          # sample pixels on the subject, like the trainer's fg sampler
sel = hit[torch.randint(0, hit.shape[0], (n_batch,), generator=g).to(dev)]
batch = rays[sel].contiguous()
target = torch.rand((n_batch, 3), generator=g).to(dev)
spp = 512
light_u = torch.rand((spp, 3), generator=g).to(dev)
shuffle_u = torch.rand((n_batch, spp), generator=g).to(dev)
params = rs.parameters() + list(mat.parameters()) + [env_base]
def step():
    for p in params: p.grad = None
    return rs.forward_backward_phys(batch, target, mat, env, spp, light_u, shuffle_u, env_base=env_base,
                                    background_color=torch.zeros(3, device=dev))
for _ in range(2): out = step()
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K): out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
lib = L.lib(); lib.start(); step(); per = lib.report()
if os.environ.get("IA_CPROFILE"):          # host side of one step: where the Python time goes (the 4096-ray step is launch-bound)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(28)
    n_launch = sum(v[0] for v in per.values())
    sys.stderr.write(f"C-ABI launches per step: {n_launch}, kernel ms (sum of event pairs): {sum(v[1] for v in per.values()):.2f}\n")
print(json.dumps(dict(workload=f"config 4: {n_batch} rays of a 540x540 frame, PBR training step (uniform_light, spp 512), fwd+bwd",
                      ms_per_step=round(dt * 1e3, 2), primary_rays_per_s=round(n_batch / dt, 1),
                      secondary_rays_per_step=out["stats"]["n_secondary"], secondary_rays_per_s=round(out["stats"]["n_secondary"] / dt, 1),
                      fg_points=out["stats"]["n_fg"], n_samples=out["n_samples"],
                      breakdown_ms={k: round(v[1], 2) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:10]})))
