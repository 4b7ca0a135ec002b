"""CPU ORACLE (test infrastructure): one worker process of bench.py's multi-core `cpu_baseline` leg.

    python -m oracle.cpu_worker <scene.npz> <rays.npy> <start> <end> <spp | 0> [<out.npz>]

loads the scene bundle written by bench.py, runs oracle/render_ref.py on rays[start:end] (relight_step when spp > 0, else
render_step) and prints the seconds the computation took (interpreter start-up and loading excluded).  With <out.npz> the
rendered maps and per-ray sample counts are kept: bench.py renders the same rays with the same random numbers on the GPU and
reports the differences (`parity_on_bench_frame`)."""
import sys
import time

import numpy as np


def load_scene(path):
    from oracle import render_ref as R
    z = np.load(path, allow_pickle=False)
    kw, lists = {}, {}
    for k in z.files:
        if "__" in k:
            name, i = k.rsplit("__", 1)
            lists.setdefault(name, {})[int(i)] = z[k]
        else:
            v = z[k]
            kw[k] = v if v.ndim else v.item()
    for name, d in lists.items():
        kw[name] = [d[i] for i in sorted(d)]
    return R.Scene(**kw)


def save_scene(path, export):
    flat = {}
    for k, v in export.items():
        if isinstance(v, (list, tuple)):
            for i, a in enumerate(v):
                flat[f"{k}__{i}"] = np.asarray(a)
        else:
            flat[k] = np.asarray(v)
    np.savez(path, **flat)


def main():
    from oracle import render_ref as R, oracle as O
    scene, rays, a, b, spp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    O.lib()
    sc = load_scene(scene)
    r = np.load(rays)[a:b]
    t0 = time.perf_counter()
    if spp > 0:
        out = R.relight_step(sc, r, spp=spp, seed=a, global_illumination=True)
    else:
        out = R.render_step(sc, r)
    dt = time.perf_counter() - t0
    if len(sys.argv) > 6:
        keep = {k: np.asarray(out[k]) for k in ("comp_rgb", "comp_normal", "opacity", "depth", "albedo", "roughness", "metallic",
                                                "comp_rgb_phys", "packed_info", "resampled_packed_info") if k in out}
        keep["n_fg"] = np.asarray(out["stats"].get("n_fg", 0))
        np.savez(sys.argv[6], **keep)
    print(f"{dt:.4f}")


if __name__ == "__main__":
    main()
