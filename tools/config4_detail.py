#!/usr/bin/env python3
"""per-launch times of the deformer search (and the other top entry points) in ONE config-4 step (4096 rays, uniform_light spp 512):
which calls are latency-bound.  python tools/config4_detail.py"""
import json, os, sys, runpy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import _lib as L
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_phys_bench.py"), run_name="not_main")
step = ns["step"]
lib = L.lib(); lib.start(); step(); det = lib.report(detail=True)
out = {}
for k, v in sorted(det.items(), key=lambda kv: -sum(c[0] for c in kv[1]))[:8]:
    out[k] = [(round(c[0], 3), c[1]) for c in v]
print(json.dumps(out))
