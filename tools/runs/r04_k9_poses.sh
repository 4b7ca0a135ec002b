#!/bin/bash
# the K9-consistent early filter against search-to-the-end + K9 on the secondary-march points of every reference pose of
# tests/golden/reference_poses.npz (4 training frames of male-3-casual, 4 out-of-distribution aist frames): one JSON line per pose
mkdir -p gpurun_out
: > gpurun_out/r04_spec_search_probe_poses.jsonl
for IA_SEED in ${IA_SEEDS:-0}; do
export IA_SEED
for pose in male-3-casual:0 male-3-casual:40 male-3-casual:80 male-3-casual:113 aist:0 aist:100 aist:200 aist:319; do
    IA_POSE=$pose IA_EPS_LIST=1e-3 timeout 600 python tools/spec_search_probe.py >> gpurun_out/r04_spec_search_probe_poses.jsonl 2>> gpurun_out/r04_spec_search_probe_poses.err
done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_spec_search_probe_poses.jsonl"):
    r = json.loads(l); s = r["spec"][0]
    print(r["pose"], "seed", r.get("seed", 0), r["points"], "mismatch", s["set_mismatch"], "lost", s["lost_root"], "sdf_max", s["sdf_max_abs"], "fetch", round(s["fetches_per_point"], 2), "/", round(r["exact"]["fetches_per_point"], 1), "ms", round(s["ms"], 2), "/", round(r["exact"]["ms"], 2), "redone", s["redone_points"])
PY
