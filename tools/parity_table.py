#!/usr/bin/env python3
"""Observed end-to-end differences of the HIP path against the reference's OWN IntrinsicAvatarModel.forward_
(tests/golden/golden_forward.npz, made by tests/golden/make_golden_forward.py from /root/reference): for every key of the
output dict of every golden run -- max / p99 / mean absolute difference per pixel (float maps) or the number of differing
entries (bool / integer keys).  The test bars of tests/test_gpu_forward_golden.py are set from this table (<= 3 x observed,
with a hard maximum); BASELINE.md / DESIGN.md section 3 quote it.

    python tools/parity_table.py > profiles/r04_parity_table.json        (prints a markdown table on stderr)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build; build.build()          # noqa: E402,E702
from tests import forward_golden as FG                           # noqa: E402

DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731
N = lambda t: t.detach().cpu().numpy()                            # noqa: E731


def stats(a, b):
    if b.dtype.kind in "biu":
        return dict(kind="discrete", entries=int(b.size), differ=int((a != b).sum()),
                    abs_diff_max=(int(np.abs(a.astype(np.int64) - b.astype(np.int64)).max()) if b.size and b.dtype.kind != "b" else None))
    e = np.abs(a.astype(np.float64) - b.astype(np.float64))
    e = e.reshape(e.shape[0], -1).max(-1) if e.ndim > 1 else e
    return dict(kind="float", entries=int(e.size), max=float(e.max()) if e.size else 0.0, p99=float(np.quantile(e, 0.99)) if e.size else 0.0,
                mean=float(e.mean()) if e.size else 0.0, ref_abs_max=float(np.abs(b).max()) if b.size else 0.0)


def run(G, tag):
    mode, spp, gi = FG.RUNS[tag]
    rs, mat, env, rays = FG.gpu_scene(G, tag)
    rnd = FG.explicit_randoms(G, tag)
    light_u = rnd["stratified_u"] if mode == "uniform_light" else rnd["light_u"]
    scatter_u = None
    if "scatter_u" in rnd:
        g = torch.Generator().manual_seed(0)
        scatter_u = torch.cat([torch.from_numpy(rnd["scatter_u"]), torch.rand((4096, 6), generator=g)]).to(DEV)
    d = rs.forward_(rays, mat, env, spp, T(light_u), T(rnd["shuffle_u"]) if "shuffle_u" in rnd else None,
                    background_color=T(G["background_color"]), global_illumination=gi, render_mode=mode, scatter_u=scatter_u)
    ref = {str(k): G[f"{tag}_out_{k}"] for k in G[tag + "_out_keys"]}
    return {k: stats(N(d[k]), ref[k]) for k in sorted(ref) if tuple(d[k].shape) == ref[k].shape}


def run_train(G):
    tag = FG.TRAIN_RUN
    rs, mat, env, rays = FG.gpu_scene(G, tag)
    rnd = FG.explicit_randoms(G, tag)
    g = torch.Generator().manual_seed(0)
    mj = torch.cat([torch.from_numpy(rnd["material_jitter"]), torch.randn((4096, 3), generator=g)]).to(DEV)
    lu = torch.cat([torch.from_numpy(rnd["light_u"]), torch.rand((4096, 3), generator=g)]).to(DEV)
    d = rs.forward_train_(rays, mat, env, 16, lu, jitter=T(rnd["near_jitter"]), material_jitter=mj, background_color=T(G["background_color"]),
                          global_illumination=True, render_mode="light")
    d.pop("stats")
    ref = {str(k): G[f"{tag}_out_{k}"] for k in G[tag + "_out_keys"]}
    return {k: stats(N(d[k]), ref[k]) for k in sorted(ref) if tuple(d[k].shape) == ref[k].shape}


def main():
    G = FG.load()
    res = {tag: run(G, tag) for tag in FG.RUNS}
    res[FG.TRAIN_RUN] = run_train(G)
    print(json.dumps(res))
    keys = sorted({k for r in res.values() for k in r})
    for k in keys:
        row = []
        for tag in res:
            s = res[tag].get(k)
            if s is None:
                row.append("-")
            elif s["kind"] == "discrete":
                row.append(f"{s['differ']}/{s['entries']}")
            else:
                row.append(f"{s['max']:.1e} / {s['p99']:.1e} / {s['mean']:.1e}")
        print(f"| {k} | " + " | ".join(row) + " |", file=sys.stderr)
    print("| key | " + " | ".join(res) + " |", file=sys.stderr)


if __name__ == "__main__":
    main()
