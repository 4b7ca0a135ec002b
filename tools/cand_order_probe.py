#!/usr/bin/env python3
"""Does the ORDER of the candidate list matter to the hash gather?  The packed list is point-major (a point's 1.27 candidates on average
are neighbours in the list, but its 2nd / 3rd candidate lies on another body part in canonical space).  Times geometry.sdf_only (12
gather launches + the SDF head) on the candidates of ~18 M march points in (a) the product order, (b) rank-major order (all first candidates
in point order, then all second ones, ...), (c) sorted by the Morton code of the CANONICAL position (the upper bound of what any
reordering could give), (d) shuffled.  python tools/cand_order_probe.py"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S, _lib as L
from tools import spec_search_probe as SP
dev = "cuda:0"
rs, rays, _ = S.build_frame(dev, 540, 540, pose="male-3-casual:0", beta=0.01, num_samples_per_ray=128)
pts = SP.march_points(rs, rays, 1 << 21)                        # sorted (posed-space Morton order), as the product evaluates them
geo, dfm = rs.geometry, rs.deformer
cand_x, _, cnt, start, Q, _, _ = dfm._candidates(pts, with_src=False, normalize=(geo.center, geo.scale))
P = pts.shape[0]
pid = torch.repeat_interleave(torch.arange(P, device=dev), cnt.long())
rank = torch.arange(Q, device=dev) - start.long()[pid]
orders = {"product (point-major)": torch.arange(Q, device=dev),
          "rank-major": torch.argsort(rank * P + pid),
          "shuffled": torch.randperm(Q, device=dev)}
# canonical-space Morton order of the candidates themselves (unit-cube coordinates -> 10 bits per axis)
origin = (C.c_float * 3)(0.0, 0.0, 0.0)
o = torch.empty(Q, dtype=torch.int32, device=dev)
nb = int(L.lib().ia_morton_order_tmp_bytes(L.i64(Q)))
tmp = torch.empty(nb, dtype=torch.uint8, device=dev)
L.check(L.lib().ia_morton_order(L.i64(Q), L.ptr(cand_x.contiguous()), origin, L.f32(1023.0), L.i32(0), L.ptr(o), L.ptr(tmp), C.c_size_t(nb), L.stream()), "order")
orders["canonical Morton"] = o.long()
res = dict(points=P, candidates=Q, per_point=round(Q / P, 3), share_by_rank=[round(float((rank == k).float().mean()), 4) for k in range(4)])
ref = None
for name, perm in orders.items():
    x = cand_x[perm].contiguous()
    for _ in range(2):
        y = geo.sdf_only(x, normalized=True)
    lib = L.lib(); lib.start()
    for _ in range(3):
        y = geo.sdf_only(x, normalized=True)
    per = lib.report()
    res[name] = {k: round(v[1] / v[0], 3) for k, v in per.items()}
    back = torch.empty_like(y); back[perm] = y
    ref = back if ref is None else ref
    assert torch.equal(back, ref), name            # the values do not depend on the order
print(json.dumps(res))
