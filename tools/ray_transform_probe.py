#!/usr/bin/env python3
"""Which summation form of o R^T + t equals the library GEMM the torch expression of SNARFDeformer.transform_rays_w2s issues, bit for bit?
(ia_transform_rays_w2s variants 0 / 1 / 2 against `rays[:, :3] @ w2s[:3, :3].T + w2s[None, :3, 3]`, random rays and rigid transforms, and
against numpy's float32 product -- what the CPU oracle computes.)    python tools/ray_transform_probe.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import _lib as L
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
res = {}
for trial in range(6):
    n = [1000, 4096, 291600, 1_000_003, 4096, 291600][trial]
    A = torch.linalg.qr(torch.randn((3, 3), generator=g))[0]
    w2s = torch.eye(4)
    w2s[:3, :3] = A
    w2s[:3, 3] = torch.randn(3, generator=g) * 2
    rays = torch.cat([torch.randn((n, 3), generator=g) * 3, torch.nn.functional.normalize(torch.randn((n, 3), generator=g), dim=-1), torch.zeros((n, 2))], 1)
    rg, wg = rays.to(dev).contiguous(), w2s.to(dev).contiguous()
    o = rg[:, :3] @ wg[:3, :3].T + wg[None, :3, 3]
    d = rg[:, 3:6] @ wg[:3, :3].T
    dist = torch.linalg.norm(o, dim=-1, keepdim=True)
    ref = torch.cat([o, d, dist - 1, dist + 1], -1)
    rn, wn = rays.numpy(), w2s.numpy()
    on = rn[:, :3] @ wn[:3, :3].T + wn[None, :3, 3]
    dn = rn[:, 3:6] @ wn[:3, :3].T
    nn = np.linalg.norm(on, axis=-1, keepdims=True)
    for v in (0, 1, 2, 4, 8):
        out = torch.empty((n, 8), device=dev)
        L.check(L.lib().ia_transform_rays_w2s(L.i64(n), L.ptr(rg), L.i32(8), L.ptr(wg), L.i32(v), L.ptr(out), L.stream()), "ia_transform_rays_w2s")
        r = res.setdefault(v, dict(vs_torch_gpu_od_mismatch=0, vs_torch_gpu_nearfar_mismatch=0, vs_numpy_od_mismatch=0, elements=0))
        r["vs_torch_gpu_od_mismatch"] += int((out[:, :6] != ref[:, :6]).sum())
        r["vs_torch_gpu_nearfar_mismatch"] += int((out[:, 6:] != ref[:, 6:]).sum())
        r["vs_numpy_od_mismatch"] += int((out[:, :6].cpu().numpy() != np.concatenate([on, dn], 1)).sum())
        r["vs_numpy_nearfar_mismatch"] = r.get("vs_numpy_nearfar_mismatch", 0) + int((out[:, 6:].cpu().numpy() != np.concatenate([nn - 1, nn + 1], 1)).sum())
        r["elements"] += n * 6
    res.setdefault("torch_gpu_vs_numpy_od_mismatch", 0)
    res["torch_gpu_vs_numpy_od_mismatch"] += int((ref[:, :6].cpu().numpy() != np.concatenate([on, dn], 1)).sum())
print(json.dumps(res))
