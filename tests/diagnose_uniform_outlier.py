#!/usr/bin/env python3
"""Diagnostic (test infrastructure -- it calls the oracle): WHY do the largest same-state radiance differences of
tests/test_gpu_relight_oracle.py::test_relight_uniform_light_mode_vs_oracle differ?

    python tests/diagnose_uniform_outlier.py [--top 6] > gpurun_out/uniform_outliers.json

Same scene, rays and uniforms as the test.  For the `top` same-state re-samples with the largest |dLo| it prints both sides' inputs to the
estimator (normal, view / light direction, n.l, n.v, n.h, roughness, metallic, albedo, secondary transmittance, indirect radiance), both
radiances, and the ORACLE's estimator evaluated on the GPU's inputs of that sample: if that reproduces the GPU's value the difference is
the inputs' (an ill-conditioned BRDF term), if it reproduces the oracle's value it is the shading kernel's."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=6)
    args = ap.parse_args()
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S, fields, pbr
    from oracle import render_ref as R, pbr_ref as Pb
    from tests.test_gpu_relight_oracle import hdri, T, N, _same_state
    rs, rays, export = S.build_frame(DEV, 32, 32, pose_seed=0, beta=0.01, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64,
                                     smooth_iters=5, hash_amp=1e-2)
    mat = fields.VolumeMaterial(seed=2).to(DEV)
    env = pbr.EnvironmentLightTensor(T(hdri()))
    env.update_pdf()
    sc = R.Scene(**export, **S.export_phys(mat, env.base))
    n, spp = rays.shape[0], 512
    rng = np.random.default_rng(7)
    light_u = rng.random((spp, 3), dtype=np.float32)
    shuffle_u = rng.random((n, spp), dtype=np.float32)
    bg = np.array([0.0, 0.0, 0.0], np.float32)
    ref = R.relight_step(sc, N(rays), spp=spp, light_u=light_u, shuffle_u=shuffle_u, global_illumination=True, background_color=bg,
                         render_mode="uniform_light")
    out = rs.relight(rays, mat, env, spp, T(light_u), T(shuffle_u), background_color=T(bg), global_illumination=True,
                     render_mode="uniform_light", return_index_lists=True)
    fg_ref = np.zeros(ref["stats"]["n_resampled"], bool); fg_ref[ref["fg_indices"]] = True
    fg_gpu = np.zeros(out["stats"]["n_resampled"], bool); fg_gpu[N(out["fg_indices"])] = True
    same = fg_ref & fg_gpu
    ig, ir = (np.cumsum(fg_gpu) - 1)[same], (np.cumsum(fg_ref) - 1)[same]
    tr_g, tr_r = N(out["secondary_tr"])[ig, 0], ref["secondary_tr"][ir, 0]
    st = _same_state(out, ref, same, ig, ir, tr_g, tr_r)
    Lo_g, Lo_r = N(out["fg_Lo"])[ig], ref["fg_Lo"][ir]
    scale = float(np.abs(Lo_r).mean() + 1e-6)
    d = np.abs(Lo_g - Lo_r).max(-1) / scale
    d_same = np.where(st, d, -1.0)
    top = np.argsort(-d_same)[:args.top]
    dirs_smpl, inv_pdf_all = Pb.uniform_sphere_stratified(16, 32, light_u[:, :2])
    rows = []
    exg = {k: N(v) for k, v in out["fg_extras"].items()}
    exr = ref["fg_extras"]
    sh = ref["shuffled"]
    for k in top:
        a, b = int(ig[k]), int(ir[k])
        ld = dirs_smpl[sh[b]]
        row = dict(rank_value=float(d[k]), same_state=bool(st[k]), Lo_gpu=Lo_g[k].tolist(), Lo_oracle=Lo_r[k].tolist(),
                   light_dir=ld.tolist(), inv_pdf=float(inv_pdf_all[sh[b]][0]))
        for side, ex, i in (("gpu", exg, a), ("oracle", exr, b)):
            nrm, view = ex["normals"][i], ex["t_dirs"][i]
            wi = -view
            h = wi + ld
            h = h / max(np.linalg.norm(h), 1e-12)
            row[side] = dict(normal=nrm.tolist(), view=view.tolist(), n_dot_l=float(nrm @ ld), n_dot_v=float(nrm @ wi), n_dot_h=float(nrm @ h),
                             roughness=float(np.ravel(ex["roughness"][i])[0]), metallic=float(np.ravel(ex["metallic"][i])[0]),
                             albedo=ex["albedo"][i].tolist())
        row["gpu"].update(tr=float(tr_g[k]), ind_rgb=N(out["secondary_rgb"])[a].tolist())
        row["oracle"].update(tr=float(tr_r[k]), ind_rgb=ref["secondary_rgb"][b].tolist())
        # the oracle's estimator on the GPU's inputs of this one sample
        one = lambda v: np.asarray(v, np.float32)[None]      # noqa: E731
        R3 = np.asarray(sc.w2s[:3, :3], np.float32)
        Lo_x, Ld_x, Ls_x, _ = Pb.pbr_uniform_light_shade(one(exg["normals"][a]), one(exg["albedo"][a]), one(np.ravel(exg["roughness"][a])[:1])[0],
                                                         one(np.ravel(exg["metallic"][a])[:1])[0], one(exg["t_dirs"][a]), one(ld),
                                                         one([tr_g[k]])[0], one(N(out["secondary_rgb"])[a]), sc.env_base, R3,
                                                         one([inv_pdf_all[sh[b]][0]])[0])
        row["oracle_estimator_on_gpu_inputs"] = Lo_x[0].tolist()
        row["that_minus_gpu_over_mean"] = float(np.abs(Lo_x[0] - Lo_g[k]).max() / scale)
        row["that_minus_oracle_over_mean"] = float(np.abs(Lo_x[0] - Lo_r[k]).max() / scale)
        rows.append(row)
    print(json.dumps(dict(mean_radiance=scale, same_state_samples=int(st.sum()), rows=rows), indent=1))


if __name__ == "__main__":
    main()
