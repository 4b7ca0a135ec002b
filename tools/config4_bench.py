#!/usr/bin/env python3
"""bench.py's `config4` object alone (the reference's 4096-ray training batch: uniform_light, spp 512, fwd + bwd + Adam) -- for A/B runs of
host-orchestration changes without the 540 x 540 headline step.    python tools/config4_bench.py [--steps 20] [--repeat 3] [--audit]"""
import argparse
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--n-batch", type=int, default=4096)
    args = ap.parse_args()
    import bench as B
    from intrinsicavatar_amd import build
    build.build()
    dev = "cuda:0"
    torch.cuda.memory._set_allocator_settings("roundup_power2_divisions:8")
    rs, rays, export, mat, sg = B.build_headline(dev, 540, 1024, 0, "male-3-casual:0")
    bg = torch.ones(3, device=dev)
    res = [B.measure_config4(rs, rays, mat, sg, dev, bg, n_batch=args.n_batch, steps=args.steps) for _ in range(args.repeat)]
    best = min(res, key=lambda r: r["ms_per_step"])
    best["ms_per_step_runs"] = [r["ms_per_step"] for r in res]
    best["env"] = {k: v for k, v in os.environ.items() if k.startswith("IA_")}
    print(json.dumps(best))


if __name__ == "__main__":
    main()
