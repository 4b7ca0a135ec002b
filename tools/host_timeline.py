#!/usr/bin/env python3
"""CPU enqueue time vs GPU time per phase of one training step (where does the host stall the GPU?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, train
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01, num_samples_per_ray=128)
n = rays.shape[0]
g = torch.Generator().manual_seed(1)
trgb = torch.rand((n, 3), generator=g).to(dev); tmask = (torch.rand(n, generator=g) > 0.5).float().to(dev)
params = rs.parameters()
def step(timing=None):
    for p in params: p.grad = None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    c = [time.perf_counter()]
    ev[0].record()
    s = rs.sample(rays, None); c.append(time.perf_counter()); ev[1].record()
    rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats = s
    out = train.shade_differentiable(rs, rays_o, rays_d, ray_indices, t_starts, t_ends, packed_info); c.append(time.perf_counter()); ev[2].record()
    loss = train.training_loss(out, trgb, tmask); c.append(time.perf_counter()); ev[3].record()
    loss.backward(); c.append(time.perf_counter()); ev[4].record()
    torch.cuda.synchronize(); c.append(time.perf_counter())
    if timing is not None:
        names = ["sample", "shade_fwd", "loss", "backward"]
        for i, nm in enumerate(names):
            print(f"{nm:10s} cpu-enqueue {1e3*(c[i+1]-c[i]):7.2f} ms   gpu-span {ev[i].elapsed_time(ev[i+1]):7.2f} ms")
        print(f"final drain {1e3*(c[5]-c[4]):.2f} ms; total wall {1e3*(c[5]-c[0]):.2f} ms")
for _ in range(3): step()
step(True); step(True)
# backward alone under the profiler: top CPU ops
from torch.profiler import profile, ProfilerActivity
for p in params: p.grad = None
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
