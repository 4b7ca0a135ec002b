#!/usr/bin/env python3
"""where a relit frame's wall time goes when the pose changes every frame (tools/relight_bench.py): repose / relight split,
device allocations of the caching allocator per frame."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, fields, pbr, _lib as L
dev = "cuda:0"; hw = 540; spp = int(os.environ.get("IA_SPP", "1024"))
mat = fields.VolumeMaterial(seed=2).to(dev)
H, W = 1024, 2048
v, u = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
img = np.where((v < 0.5)[..., None], np.stack([0.3 + 0.4 * (1 - v), 0.4 + 0.4 * (1 - v), 0.6 + 0.4 * (1 - v)], -1), 0.08)
env = pbr.EnvironmentLightTensor(torch.from_numpy(img.astype(np.float32)).to(dev)); env.update_pdf()
light_u = torch.rand((spp, 3)).to(dev)
rs, rays, _ = S.build_frame(dev, hw, hw, pose="aist:0", beta=0.01)
out = []
for k, pose in enumerate(["aist:0", "aist:100", "aist:200", "aist:319", "aist:0", "aist:100"]):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S.repose(rs, pose)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    su = torch.rand((rays.shape[0], spp), device=dev)
    lib = L.lib(); lib.start()
    o = rs.relight(rays, mat, env, spp, light_u, su, global_illumination=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    kern = sum(v[1] for v in lib.report().values())
    st1 = torch.cuda.memory_stats()
    out.append(dict(pose=pose, repose_s=round(t1 - t0, 3), relight_s=round(t2 - t1, 3), abi_kernel_s=round(kern / 1e3, 3),
                    device_allocs=st1["num_device_alloc"] - st0["num_device_alloc"], device_frees=st1["num_device_free"] - st0["num_device_free"],
                    alloc_retries=st1["num_alloc_retries"] - st0["num_alloc_retries"],
                    reserved_GiB=round(st1["reserved_bytes.all.current"] / 2**30, 1)))
print(json.dumps(out, indent=0))
