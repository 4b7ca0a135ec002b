#!/usr/bin/env python3
"""Per-stage CPU (oracle, 1 thread, ray subsample, linearly extrapolated) vs MI355X timing table for the forward
render_step at 540x540.  Prints markdown."""
import os, sys, time, functools
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, _lib as L
from oracle import oracle as O, render_ref as R

rs, rays, export = S.build_frame("cuda:0", 540, 540, pose_seed=0, beta=0.01)
n = rays.shape[0]
for _ in range(2):
    rs.forward(rays)
torch.cuda.synchronize()
lib = L.lib(); lib.start()
rs.forward(rays)
gpu = lib.report()
# ---- CPU: every 71st ray (4107 rays), per-oracle-function wall time
acc = {}
def timed(name, fn):
    @functools.wraps(fn)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
for name in ("traverse_grids", "fuse_broyden", "filter", "sdf_field", "hashgrid_fwd", "sh4", "mlp_fwd", "laplace_alpha",
             "render_weight_from_alpha", "ray_resampling_merge", "unpack_info", "pack_info", "accumulate_along_rays"):
    setattr(O, name, timed(name, getattr(O, name)))
stride = 71
sample = rays[::stride].cpu().numpy()
t0 = time.perf_counter(); R.render_step(R.Scene(**export), sample); tcpu = time.perf_counter() - t0
scale = n / sample.shape[0]
rows = [
    ("occupancy-grid marching (T1)", ["traverse_grids"], ["ia_traverse_grids_count", "ia_exclusive_scan_i64", "ia_traverse_grids_fill", "ia_traverse_grids_fused"]),
    ("Broyden root search (K8)", ["fuse_broyden"], ["ia_fuse_broyden"]),
    ("candidate filter / compaction / select (K9 + glue)", ["filter"], ["ia_deform_filter_count", "ia_deform_compact", "ia_deform_select", "ia_exclusive_scan_i32"]),
    ("hash-grid encode + SDF/radiance MLPs (T4, T5, F1, F2)", ["sdf_field", "hashgrid_fwd", "sh4", "mlp_fwd"], ["ia_hashgrid_fwd", "ia_hashgrid_fwd_xcd", "ia_sh4_fwd", "ia_mlp_fwd"]),
    ("alpha / transmittance weights / accumulation (T2, T3)", ["laplace_alpha", "render_weight_from_alpha", "accumulate_along_rays"], ["ia_laplace_alpha", "ia_render_weight_from_alpha", "ia_accumulate_along_rays", "ia_ray_points", "ia_shade_prep"]),
    ("importance resampling + pack/unpack (K2, K5, pack_info)", ["ray_resampling_merge", "unpack_info", "pack_info"], ["ia_resample_packed_info", "ia_ray_resampling_merge", "ia_unpack_info", "ia_pack_info"]),
]
print(f"| stage | CPU oracle, {sample.shape[0]} rays, 1 thread [ms] | CPU extrapolated to {n} rays [s] | MI355X, full frame [ms] | ratio |")
print("|---|---|---|---|---|")
tc, tg = 0.0, 0.0
for title, cpu_fns, gpu_fns in rows:
    c = sum(acc.get(f, 0.0) for f in cpu_fns); g = sum(gpu.get(f, (0, 0.0))[1] for f in gpu_fns)
    tc += c; tg += g
    print(f"| {title} | {c * 1e3:.0f} | {c * scale:.1f} | {g:.2f} | {c * scale * 1e3 / max(g, 1e-9):.0f}x |")
print(f"| **all stages** (CPU total incl. numpy glue {tcpu * 1e3:.0f} ms) | {tc * 1e3:.0f} | {tcpu * scale:.1f} | {tg:.2f} (kernels) | {tcpu * scale * 1e3 / tg:.0f}x |")
print(f"\nhost cores on this box: {os.cpu_count()} (the oracle uses 1)")
