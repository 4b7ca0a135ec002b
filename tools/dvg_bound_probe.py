#!/usr/bin/env python3
"""Could a search that DIVERGES at its second fetch be recognised without that fetch?  (Design probe, no kernel.)
g(x) + xd = sum_k w_k(x) (A_k x + b_k) is a convex combination of the 8 corner maps of x's voxel cell, so with the cell's mean map
(Abar, bbar) and the radius  r_c = max_k ( |A_k - Abar|_F h + |(A_k - Abar) c0 + b_k - bbar| )  (c0 cell centre, h half its diagonal)
      |g(x)| >= |Abar x + bbar - xd| - r_c      for every x in the cell:
52 bytes per cell instead of the fetch's 384.  The exact search is emulated (tools/k9_rule_probe.py: whole trajectories); per fetch
index k the probe counts the searches that end there by divergence and how many of them the bound DECIDES (lower bound > 0.1 m + 1e-3),
and the fetches at which the bound would be evaluated in vain.    python tools/dvg_bound_probe.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k9_rule_probe as K
import spec_search_probe as SP
from intrinsicavatar_amd import synthetic as S

rs, rays, _ = S.build_frame(SP.dev, 540, 540, pose_seed=0, beta=0.01, pose=os.environ.get("IA_POSE", "male-3-casual:0"))
xd = SP.march_points(rs, rays, int(os.environ.get("IA_NSEC", str(1 << 19))))
dfm = rs.deformer
vJ, tfs, offk, sck, bones = dfm.voxel_J_cl[0], dfm.tfs[0], dfm.offset_kernel, dfm.scale_kernel, dfm.init_bones
D, H, W, _ = vJ.shape
dev = vJ.device
sc3, of3 = sck.reshape(-1)[:3], offk.reshape(-1)[:3]
# ---- per-cell table: mean map + radius
corners = []
for c in range(8):
    cx, cy, cz = c & 1, (c >> 1) & 1, (c >> 2) & 1
    corners.append(vJ[cz:D - 1 + cz, cy:H - 1 + cy, cx:W - 1 + cx].reshape(D - 1, H - 1, W - 1, 3, 4))
C = torch.stack(corners, 0)                                       # [8, D-1, H-1, W-1, 3, 4]
M = C.mean(0)
iz, iy, ix = torch.meshgrid(torch.arange(D - 1, device=dev), torch.arange(H - 1, device=dev), torch.arange(W - 1, device=dev), indexing="ij")
g0 = torch.stack([(ix.float() + 0.5) / (W - 1) * 2 - 1, (iy.float() + 0.5) / (H - 1) * 2 - 1, (iz.float() + 0.5) / (D - 1) * 2 - 1], -1)
c0 = g0 / sc3 - of3                                              # cell centres, canonical metres
half = 0.5 * torch.stack([2 / ((W - 1) * sc3[0].abs()), 2 / ((H - 1) * sc3[1].abs()), 2 / ((D - 1) * sc3[2].abs())]).norm()
dA = C[..., :3] - M[None, ..., :3]
db = C[..., 3] - M[None, ..., 3]
rad = (dA.reshape(8, D - 1, H - 1, W - 1, 9).norm(dim=-1) * half + (torch.einsum("kzyxij,zyxj->kzyxi", dA, c0) + db).norm(dim=-1)).amax(0)
tab_stats = dict(cells=int(rad.numel()), radius_p50=float(rad.flatten().median()), radius_p90=float(rad.flatten().kthvalue(int(0.9 * rad.numel())).values),
                 radius_p99=float(rad.flatten().kthvalue(int(0.99 * rad.numel())).values), half_diag_m=float(half))
chunk = 1 << 19
acc = dict(points=0, searches=0, fetches=0)
by_k = {k: dict(end_diverged=0, decided=0, evaluated_in_vain=0, at_fetch=0) for k in (1, 2, 3)}
for a in range(0, xd.shape[0], chunk):
    x = xd[a:a + chunk]
    traj, nfetch, valid, xfin, jn, jtraj = K.search(x, vJ, tfs, bones, offk, sck)
    gtraj = K.search.gtraj
    P, I = valid.shape
    acc["points"] += P; acc["searches"] += P * I; acc["fetches"] += int(nfetch.sum())
    for k in (1, 2, 3):
        at = nfetch > k                                           # the search issues fetch k (0-based)
        xk = traj[:, :, k]
        ends_div = at & (nfetch == k + 1) & ~(gtraj[:, :, k] <= K.DVG) & ~(gtraj[:, :, k] < K.CVG)
        g = (xk + offk) * sck
        fx, fy, fz = ((g[..., 0] + 1) / 2 * (W - 1)).floor(), ((g[..., 1] + 1) / 2 * (H - 1)).floor(), ((g[..., 2] + 1) / 2 * (D - 1)).floor()
        inside = at & torch.isfinite(fx) & (fx >= 0) & (fx < W - 1) & (fy >= 0) & (fy < H - 1) & (fz >= 0) & (fz < D - 1)
        cxx, cyy, czz = fx.clamp(0, W - 2).nan_to_num(0).long(), fy.clamp(0, H - 2).nan_to_num(0).long(), fz.clamp(0, D - 2).nan_to_num(0).long()
        Mm = M[czz, cyy, cxx]                                     # [P,I,3,4]
        lb = (torch.einsum("pnij,pnj->pni", Mm[..., :3], xk.nan_to_num(0)) + Mm[..., 3] - x[:, None, :]).norm(dim=-1) - rad[czz, cyy, cxx]
        decided = inside & (lb > K.DVG + 1e-3)
        assert not bool((decided & at & (gtraj[:, :, k] <= K.DVG)).any()), "the bound must never decide a search that does not diverge there"
        by_k[k]["at_fetch"] += int(at.sum())
        by_k[k]["end_diverged"] += int(ends_div.sum())
        by_k[k]["decided"] += int((decided & ends_div).sum())
        by_k[k]["evaluated_in_vain"] += int((at & ~decided).sum())
print(json.dumps(dict(table=tab_stats, **acc, fetches_per_point=round(acc["fetches"] / acc["points"], 2), by_fetch_index=by_k)))
