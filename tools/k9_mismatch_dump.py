#!/usr/bin/env python3
"""the points whose post-K9 candidate set differs between search-to-the-end and the K9-consistent early filter (tools/spec_search_probe.py
counts them): per init the exact search's (valid, kept, root), the filtered search's (completed valid, root) and the distances that decide
K9 -- which retired search would have been kept, how far from the root that retired it.  IA_POSE selects the pose."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import spec_search_probe as P
from intrinsicavatar_amd import synthetic as S, fast_snarf

dev = "cuda:0"
pose = os.environ.get("IA_POSE", "aist:319")
rs, rays, _ = S.build_frame(dev, 540, 540, pose_seed=0, beta=0.01, pose=pose)
pts = P.march_points(rs, rays, int(os.environ.get("IA_NSEC", str(1 << 21))))
dfm = rs.deformer
os.environ["IA_BROYDEN_SCHEDULE"] = "persistent"
n = pts.shape[0]
I = dfm.init_bones.shape[0]
x0 = torch.zeros((1, n, I, 3), device=dev); v0 = torch.zeros((1, n, I), dtype=torch.bool, device=dev)
Ji = torch.zeros((1, n, I, 3, 3), device=dev)
vj = fast_snarf.ChannelLastVoxelJ(dfm.voxel_J_cl)
fast_snarf.fuse_broyden(x0, pts.reshape(1, n, 3), None, vj, dfm.tfs, dfm.init_bones, True, Ji, v0, dfm.offset_kernel, dfm.scale_kernel, 1e-5, 1e-1)
k0 = fast_snarf.filter(x0, v0)
x1, v1 = P.search(dfm, pts, 1e-3)
k1 = fast_snarf.filter(x1, v1)
bad = torch.nonzero((k0 != k1).any(-1)[0])[:, 0]
out = []
for p in bad.tolist():
    xe, ve, ke, xs, vs, ks = x0[0, p], v0[0, p], k0[0, p], x1[0, p], v1[0, p], k1[0, p]
    jn = Ji[0, p].reshape(I, 9).norm(dim=-1)
    rec = dict(point=p, xd=pts[p].tolist(), inits=[])
    for i in range(I):
        if not bool(ve[i]):
            continue
        later = [j for j in range(i + 1, I) if bool(ve[j])]
        d_later = [float((xe[i] - xe[j]).norm()) for j in later]
        rec["inits"].append(dict(init=i, exact_kept=bool(ke[i]), spec_completed=bool(vs[i]), spec_kept=bool(ks[i]), root=xe[i].tolist(),
                                 jinv_fro=float(jn[i]), dist_to_later_valid={j: d for j, d in zip(later, d_later)}))
    out.append(rec)
print(json.dumps(dict(pose=pose, points=n, mismatching=len(out), detail=out)))

# ---- where along its trajectory would each wrongly retired search have been retired?  (torch emulation of the exact search with the
# whole trajectory kept: tools/k9_rule_probe.py)
if os.environ.get("IA_DUMP_TRAJ", "1") == "1" and len(out) > 0:
    import k9_rule_probe as K
    xd = pts[bad]
    vJ = dfm.voxel_J_cl[0]
    offk, sck = dfm.offset_kernel, dfm.scale_kernel
    traj, nfetch, valid, xfin, jn, jtraj = K.search(xd, vJ, dfm.tfs[0], dfm.init_bones, offk, sck)
    dims = vJ.shape[:3]
    keep = K.k9(xfin, valid)
    lines = []
    for q, p in enumerate(bad.tolist()):
        ke, ks = k0[0, p], k1[0, p]
        for i in range(I):
            if not (bool(ke[i]) and not bool(ks[i])):
                continue                                   # an exact survivor the filtered search does not have
            later = [j for j in range(i + 1, I) if bool(ks[j])]          # roots the filtered search had recorded by then
            for k in range(int(nfetch[q, i])):
                xk = traj[q, i, k]
                rows = []
                for j in later:
                    dinf = float((xk - x1[0, p, j]).abs().max())
                    if dinf < 2e-3:
                        same = int(K.cell_id(xk, offk, sck, dims) == K.cell_id(x1[0, p, j], offk, sck, dims))
                        rows.append((j, round(dinf, 6), same, round(float(Ji[0, p, j].reshape(9).norm()), 2)))
                if rows:
                    lines.append(dict(point=p, init=i, k=k, of=int(nfetch[q, i]), own_jinv=round(float(jtraj[q, i, k]), 2),
                                      final_jinv=round(float(jn[q, i]), 2), emul_valid=bool(valid[q, i]), near=rows))
    print(json.dumps(dict(trajectory_hits=lines)))
