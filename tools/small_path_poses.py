#!/usr/bin/env python3
"""The small-batch search (one lane per (point, init), K9 literally: broyden_items_rows_kernel + rows_flagged_kernel) against the
early-filter search on the eight committed frames of the reference's pose files: the march points of a 540 x 540 frame per pose, in
batches of 2^18 points, through SNARFDeformer._candidates both ways (IA_BR_SMALL_MAX = 0 / default).  Per pose: points compared, points
whose candidate count differs, batches whose packed lists (positions, source inits) are not bit-identical.  Both must be 0 wherever the
early filter is K9-consistent."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()
from intrinsicavatar_amd import synthetic as S
from tools import spec_search_probe as SP
dev = "cuda:0"
B = 1 << 18
tot = dict(points=0, count_differs=0, batches=0, batches_not_identical=0)
for pose in ("male-3-casual:0", "male-3-casual:40", "male-3-casual:80", "male-3-casual:113", "aist:0", "aist:100", "aist:200", "aist:319"):
    rs, rays, _ = S.build_frame(dev, 540, 540, pose=pose, beta=0.01, num_samples_per_ray=128)
    pts = SP.march_points(rs, rays, int(os.environ.get("IA_NSEC", str(1 << 19))))
    dfm = rs.deformer
    row = dict(pose=pose, points=0, count_differs=0, batches=0, batches_not_identical=0)
    for a in range(0, pts.shape[0], B):
        sub = pts[a:a + B].contiguous()
        os.environ["IA_BR_SMALL_MAX"] = "0"
        want = dfm._candidates(sub, with_src=True)
        del os.environ["IA_BR_SMALL_MAX"]
        got = dfm._candidates(sub, with_src=True)
        assert dfm._tls.n_over == 0
        row["points"] += sub.shape[0]
        row["count_differs"] += int((got[2] != want[2]).sum())
        row["batches"] += 1
        same = got[4] == want[4] and torch.equal(got[0].view(torch.int32), want[0].view(torch.int32)) and torch.equal(got[1], want[1])
        row["batches_not_identical"] += 0 if same else 1
    print(json.dumps(row), flush=True)
    for k in tot:
        tot[k] += row[k]
    del rs, pts
    torch.cuda.empty_cache()
print(json.dumps(dict(total=tot)))
