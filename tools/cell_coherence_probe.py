import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tools import spec_search_probe as SP
from intrinsicavatar_amd import synthetic as S
rs, rays, _ = S.build_frame(SP.dev, 540, 540, pose_seed=0, beta=0.01, pose="male-3-casual:0")
pts = SP.march_points(rs, rays, 1 << 21)
dfm = rs.deformer
for eps in (1e-3,):
    cnt = torch.zeros(5, dtype=torch.int64, device=SP.dev)
    SP.search(dfm, pts, eps, counters=cnt)
    c = cnt.tolist()
    print(json.dumps(dict(points=int(pts.shape[0]), fetches=c[0], in_leader_cell=round(c[1]/c[0],4), in_second_cell=round(c[2]/c[0],4), lane_slots_per_fetch=round(c[4]/c[0],4))))
