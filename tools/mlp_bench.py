#!/usr/bin/env python3
"""MFMA utilisation of the fused MLP kernels (ia_mlp_fwd / ia_mlp_bwd / ia_sdf_mlp_bwd) at render_step batch sizes.
useful FLOPs = 2 x MACs of the reference layers (SDF 3072, radiance 8576, material 7488 MACs / point;
the SDF-normal path adds the 64x35 g_h GEMM), peak = 157.3 TFLOP/s fp32 MFMA."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intrinsicavatar_amd import build  # noqa: E402

build.build()
from intrinsicavatar_amd import fields  # noqa: E402

DEV = "cuda:0"
PEAK = 157.3


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    n = int(os.environ.get("IA_MLP_N", str(1 << 22)))
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(DEV)      # noqa: E731
    enc, xyz, feat, sh, nrm = r(n, 32), r(n, 3), r(n, 13), r(n, 16), r(n, 3)
    jac = r(n, 32, 3)
    res = {}
    cases = {
        "sdf": (0, [(enc, 32, 1, 0), (xyz, 3, 1, 0)], (r(64, 35), r(64), None, None, r(13, 64), r(13)), 13, 3072, {}),
        "sdf+normal": (0, [(enc, 32, 1, 0), (xyz, 3, 1, 0)], (r(64, 35), r(64), None, None, r(13, 64), r(13)), 13,
                       3072 + 64 * 35, dict(jac=jac, xyz_col=32, inv_scale=(0.4, 0.4, 0.4), want_grad=True)),
        "radiance": (1, [(enc, 32, 1, 0), (xyz, 3, 1, 0), (feat, 13, 1, 0), (sh, 16, 1, 0), (nrm, 3, 1, 0)],
                     (r(64, 67), r(64), r(64, 64), r(64), r(3, 64), r(3)), 3, 8576, {}),
        "material": (2, [(enc, 32, 1, 0), (xyz, 3, 1, 0), (feat, 13, 1, 0)],
                     (r(64, 48), r(64), r(64, 64), r(64), r(5, 64), r(5)), 5, 7488, {}),
    }
    for name, (kind, segs, w, out, macs, kw) in cases.items():
        ms = timeit(lambda: fields.mlp_forward(kind, segs, *w, out, **kw))
        tf = 2 * macs * n / ms / 1e9
        res[name] = dict(n=n, ms=round(ms, 3), tflops=round(tf, 1), mfma_util=round(tf / PEAK, 3),
                         gpts_per_s=round(n / ms / 1e6, 2))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
