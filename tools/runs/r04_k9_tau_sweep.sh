#!/bin/bash
# the tightness thresholds of the K9-consistent filter (snarf.hip SPEC_TAU / SPEC_TAU_SELF) on the reference poses: builds made with
# tools/ab_build.sh <name> intrinsicavatar_amd/csrc/snarf.hip -DIA_SPEC_TAU=..., one JSON line per (build, pose)
mkdir -p gpurun_out
: > gpurun_out/r04_k9_tau_sweep.jsonl
for lib in "$@"; do
  for pose in male-3-casual:0 male-3-casual:80 male-3-casual:113 aist:0 aist:200 aist:319; do
    echo "{\"lib\": \"$lib\"}" >> gpurun_out/r04_k9_tau_sweep.jsonl
    IA_AMD_LIB=$PWD/intrinsicavatar_amd/_ab/libia_amd_$lib.so IA_POSE=$pose IA_EPS_LIST=1e-3 timeout 600 python tools/spec_search_probe.py >> gpurun_out/r04_k9_tau_sweep.jsonl 2>> gpurun_out/r04_k9_tau_sweep.err
  done
done
python - <<'PY'
import json
lib = None
for l in open("gpurun_out/r04_k9_tau_sweep.jsonl"):
    r = json.loads(l)
    if "lib" in r and "pose" not in r:
        lib = r["lib"]; continue
    s = r["spec"][0]
    print(lib, r["pose"], r["points"], "mismatch", round(s["set_mismatch"] * r["points"]), "lost", round(s["lost_root"] * r["points"]), "sdf_max", s["sdf_max_abs"], "fetch", round(s["fetches_per_point"], 2), "ms", round(s["ms"], 2), "redone", s["redone_points"])
PY
