#!/bin/bash
# the K9-consistent filter WITH the cell-tightness veto at larger retirement boxes (eps) on every reference pose: kernel, not emulation
mkdir -p gpurun_out
: > gpurun_out/r04_spec_search_probe_eps.jsonl
for pose in male-3-casual:0 male-3-casual:40 male-3-casual:80 male-3-casual:113 aist:0 aist:100 aist:200 aist:319; do
    IA_POSE=$pose IA_EPS_LIST=${IA_EPS_LIST:-2e-3,3e-3,5e-3} timeout 900 python tools/spec_search_probe.py >> gpurun_out/r04_spec_search_probe_eps.jsonl 2>> gpurun_out/r04_spec_search_probe_eps.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_spec_search_probe_eps.jsonl"):
    r = json.loads(l)
    for s in r["spec"]:
        print(r["pose"], r["points"], "eps", s["eps"], "mismatch", round(s["set_mismatch"] * r["points"]), "lost", round(s["lost_root"] * r["points"]), "sdf_max", s["sdf_max_abs"],
              "fetch", round(s["fetches_per_point"], 2), "/", round(r["exact"]["fetches_per_point"], 1), "ms", round(s["ms"], 2), "/", round(r["exact"]["ms"], 2), "redone", s["redone_points"])
PY
