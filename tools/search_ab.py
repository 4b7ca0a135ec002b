#!/usr/bin/env python3
"""Same-box A/B of builds of the deformer search (tools/ab_build.sh): for every library given, a fresh process loads it through
IA_AMD_LIB, builds the headline frame's march points (deterministic) and times the search entry points on them.

    python tools/search_ab.py intrinsicavatar_amd/_ab/libia_amd_a.so intrinsicavatar_amd/_ab/libia_amd_b.so ...      (prints JSON lines)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from tools import spec_search_probe as SP
from intrinsicavatar_amd import synthetic as S
rs, rays, _ = S.build_frame(SP.dev, 540, 540, pose_seed=0, beta=0.01, pose=os.environ.get("IA_POSE", "male-3-casual:0"))
pts = SP.march_points(rs, rays, int(os.environ.get("IA_NSEC", str(1 << 21))))
dfm, geo = rs.deformer, rs.geometry
out = dict(lib=os.environ.get("IA_AMD_LIB"), points=int(pts.shape[0]))
for eps in (1e-3, 0.0):
    cnt = torch.zeros(5, dtype=torch.int64, device=SP.dev)
    SP.search(dfm, pts, eps, counters=cnt)
    out[f"fetches_per_point_eps{eps:g}"] = round(cnt[0].item() / pts.shape[0], 3)
    out[f"counter4_per_fetch_eps{eps:g}"] = round(cnt[4].item() / max(cnt[0].item(), 1), 4)      # diagnostic builds: 64 / active lanes, averaged over the fetches
    out[f"search_ms_eps{eps:g}"] = round(SP.timed(lambda: SP.search(dfm, pts, eps), reps=5), 3)
dfm.spec_eps = 1e-3
out["deform_sdf_ms_rows_eps0.001"] = round(SP.timed(lambda: dfm.deform_sdf(pts, geo), reps=5), 3)
out["candidates_rows_ms_eps0.001"] = round(SP.timed(lambda: dfm._candidates(pts, with_src=False), reps=5), 3)
print(json.dumps(out))
''' % ROOT


def main():
    for lib in sys.argv[1:]:
        env = dict(os.environ, IA_AMD_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(line[0] if line else json.dumps(dict(lib=lib, error=p.stderr[-400:])), flush=True)


if __name__ == "__main__":
    main()
