import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 48, 48, pose_seed=1, beta=0.05, num_samples_per_ray=32, grid_D=16, grid_H=64, grid_W=64, smooth_iters=3, hash_amp=3e-2)
with torch.no_grad():
    for p in rs.radiance.network.parameters(): p.add_(torch.randn_like(p) * 0.05)
    for p in rs.geometry.network.parameters(): p.add_(torch.randn_like(p) * 0.03)
n = rays.shape[0]
g = torch.Generator().manual_seed(0)
target = torch.rand((n, 3), generator=g).cuda(); tmask = (torch.rand(n, generator=g) > 0.5).float().cuda()
names = [k for k, p in list(rs.geometry.named_parameters()) + list(rs.radiance.named_parameters()) + list(rs.density.named_parameters()) if p.requires_grad]
runs = []
for it in range(4):
    for p in rs.parameters(): p.grad = None
    out = rs.forward_backward(rays, target, tmask)
    runs.append([p.grad.clone() if p.grad is not None else None for p in rs.parameters()])
    junk = torch.full((32 << 20,), float("nan") if it % 2 else 1e30, device=dev); del junk
for i, nm in enumerate(names):
    a = runs[0][i]
    if a is None: continue
    worst = max(float((r[i] - a).abs().max()) for r in runs[1:])
    print(f"{nm:50s} max|g| {float(a.abs().max()):.3e}  run-to-run max diff {worst:.3e}  rel {worst / (float(a.abs().max()) + 1e-30):.2e}")
