#!/usr/bin/env python3
"""How much of the K1-K10 golden fixtures depends on fused multiply-add contraction?  (the fixtures are the
reference's kernel bodies compiled WITHOUT contraction -- source semantics -- while nvcc's default --fmad=true contracts.)

    python tools/fma_contraction_sensitivity.py [--out profiles/r06_fma_contraction_sensitivity.json]        (build container only)

Runs tests/golden/make_golden.py a second time with IA_GOLDEN_CONTRACT=fast (-ffp-contract=fast -mfma: gcc fuses a * b + c, also across
statements, like nvcc's fmad) into a scratch directory and compares every array of golden_resampling / golden_pack / golden_snarf with the
committed contraction-free fixture: integer / boolean outputs element by element (how many differ), floats by max |d| and max relative.
Not a pin of the nvcc binary (its contraction choices are its own) -- a measurement of how far contraction CAN move these kernels'
outputs on the fixtures' inputs."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--scratch", default=os.path.join(ROOT, "tools", "scratch", "golden_fma"))
    args = ap.parse_args()
    os.makedirs(args.scratch, exist_ok=True)
    env = dict(os.environ, IA_GOLDEN_CONTRACT="fast", IA_GOLDEN_OUT=args.scratch)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py")], env=env, stdout=subprocess.DEVNULL)
    res = {}
    for f in ("golden_resampling.npz", "golden_pack.npz", "golden_snarf.npz"):
        a, b = np.load(os.path.join(ROOT, "tests", "golden", f)), np.load(os.path.join(args.scratch, f))
        rows = {}
        for k in a.files:
            x, y = a[k], b[k]
            if x.shape != y.shape:
                rows[k] = dict(shape_off=[list(x.shape), list(y.shape)])
                continue
            if x.dtype.kind in "iub":
                n = int((x != y).sum())
                if n:
                    rows[k] = dict(kind="discrete", elements=int(x.size), differ=n)
            elif x.dtype.kind == "f":
                fin = np.isfinite(x) & np.isfinite(y)
                d = np.abs(x.astype(np.float64) - y.astype(np.float64))[fin]
                n = int((x != y).sum())
                if n:
                    rel = d / np.maximum(np.abs(x.astype(np.float64))[fin], 1e-30)
                    rows[k] = dict(kind="float", elements=int(x.size), differ=n, max_abs=float(d.max(initial=0.0)),
                                   max_rel=float(rel.max(initial=0.0)), non_finite_mismatch=int((np.isfinite(x) != np.isfinite(y)).sum()))
        res[f] = dict(arrays=len(a.files), arrays_identical=len(a.files) - len(rows), arrays_that_differ=rows)
    out = dict(flags="-O2 -ffp-contract=fast -mfma against the committed -ffp-contract=off fixtures", fixtures=res)
    txt = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
