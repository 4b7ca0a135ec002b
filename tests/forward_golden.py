"""Helpers for tests/golden/golden_forward.npz (the reference's own IntrinsicAvatarModel.forward_, run on CPU by
tests/golden/make_golden_forward.py): rebuild the scene of the fixture for the CPU oracle (numpy) and for the HIP path."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RUNS = dict(light_16_nogi=("light", 16, False), light_64_gi=("light", 64, True), uniform_light_512_gi=("uniform_light", 512, True),
            mis_16_gi=("mis", 16, True), mats_16_gi=("mats", 16, True))
TRAIN_RUN = "light_16_gi_train"
UNIFORM_TRAIN_RUN = "uniform_light_512_gi_train"      # the estimator the reference ships for training (configs/config.yaml:46-48), train() mode


def load():
    return np.load(os.path.join(HERE, "golden", "golden_forward.npz"))


def hash_tables():
    """the two tables of the golden scene: closed-form in the entry index (make_golden_forward.py init_params)."""
    from intrinsicavatar_amd import synthetic as S
    n = 12599920                  # floats: 6 299 960 entries x 2 features (tiny-cuda-nn's flat `params`)
    return S.hash_table_values(n, 11, 1e-2), S.hash_table_values(n, 12, 1e-2)


def rng_draws(G, tag):
    """[(kind, array)] in the order the reference drew them."""
    kinds = [str(k) for k in G[tag + "_rng_kinds"]]
    out = []
    for i, k in enumerate(kinds):
        a = G[f"{tag}_rng_{i}"]
        if k == "rand_formula":                                    # large uniform tensors are closed-form in the index (not stored)
            k, a = "rand", formula_uniforms(a[0], a[1])
        out.append((k, a))
    return out


def formula_uniforms(seed, n):
    from intrinsicavatar_amd import synthetic as S
    return ((S.hash_table_values(int(n), int(seed), 1.0).astype(np.float64) + 1.0) / 2.0).astype(np.float32)


def explicit_randoms(G, tag):
    """the random tensors of one run under the names the oracle / the package take them:
    occ_jitter [64^3, 3, 3]; light_u; shuffle_u; stratified_u [512, 2]; scatter_u [F, 6]; train: near_jitter, material_jitter."""
    d = rng_draws(G, tag)
    out = {}
    k0, a0 = d[0]
    assert k0 == "rand_like_formula"
    out["occ_jitter"] = formula_uniforms(a0[0], a0[1])            # flat: [64^3 * 3 * 3] (test grid, 3 points per voxel) or [64^3 * 3] (training grid)
    rest = d[1:]
    mode = tag.split("_")[0] if not tag.startswith("uniform") else "uniform_light"
    if tag.endswith("_train") and mode == "uniform_light":
        # stratified near-plane jitter, material jitter, the [n_rays, 512] shuffle uniforms (:1399), the stratified sphere jitter (:680-689)
        (ka, near), (kb, mj), (kc, sh), (kd, su) = rest
        assert (ka, kb, kc, kd) == ("rand_like", "randn_like", "rand", "emitter.sample_uniform_sphere_stratified")
        out.update(near_jitter=near, material_jitter=mj, shuffle_u=sh.reshape(G["rays"].shape[0], -1), stratified_u=su)
        return out
    if tag.endswith("_train"):
        # stratified near-plane jitter, material jitter, the (unused) shuffle draw, emitter.sample(F)
        (ka, near), (kb, mj), (kc, _), (kd, lu) = rest
        assert (ka, kb, kc, kd) == ("rand_like", "randn_like", "rand", "emitter.sample")
        out.update(near_jitter=near, material_jitter=mj, light_u=lu)
        return out
    assert rest[0][0] == "emitter.sample"                      # prepare(): self.secondary_rays_d = emitter.sample(spp)
    out["light_u"] = rest[0][1]
    if mode in ("light", "uniform_light"):
        assert rest[1][0] == "rand"
        out["shuffle_u"] = rest[1][1].reshape(G["rays"].shape[0], -1)
        if mode == "uniform_light":
            assert rest[2][0] == "emitter.sample_uniform_sphere_stratified"
            out["stratified_u"] = rest[2][1]
    elif mode == "mats":
        assert rest[1][0] == "scatterer.sample"
        out["scatter_u"] = np.concatenate([rest[1][1], np.zeros_like(rest[1][1])], 1)
    elif mode == "mis":
        assert rest[1][0] == "scatterer.sample" and rest[2][0] == "emitter.sample"
        out["scatter_u"] = np.concatenate([rest[1][1], rest[2][1]], 1)
    return out


def oracle_scene(G, tag):
    """oracle/render_ref.Scene of the golden scene with run `tag`'s occupancy grid."""
    from oracle import render_ref as R
    st = lambda k: G["state_" + k]      # noqa: E731
    t_geo, t_rad = hash_tables()

    def wn(g, v):
        return (g * v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    bbox = G["rig_cano_bbox"]
    center, scale = ((bbox[0] + bbox[1]) / 2).astype(np.float32), (bbox[1] - bbox[0]).astype(np.float32)
    Ws, bs = [], []
    for i in range(3):
        w = st(f"material.network.weights_per_layer.{i}").astype(np.float64)
        c = np.log1p(np.exp(st(f"material.network.lipshitz_bound_per_layer.{i}").astype(np.float64)))
        Ws.append((w * np.minimum(c / np.abs(w).sum(1), 1.0)[:, None]).astype(np.float32))
        bs.append(st(f"material.network.biases_per_layer.{i}"))
    return R.Scene(
        w2s=G["rig_w2s"], tfs=G["rig_tfs"], voxel_J=G["rig_ref_voxel_J"], offset_kernel=G["rig_offset_kernel"], scale_kernel=G["rig_scale_kernel"],
        binaries=G[tag + "_occ_binaries"][0], aabb=G[tag + "_occ_aabb"][0], step=float(G["render_step_size"]), beta=float(abs(st("density.beta")) + 1e-4),
        geo_center=center, geo_scale=scale, geo_params=t_geo, geo_mask=np.ones(32, np.float32),
        geo_W1=wn(st("geometry.network.layers.0.weight_g"), st("geometry.network.layers.0.weight_v")), geo_b1=st("geometry.network.layers.0.bias"),
        geo_W2=wn(st("geometry.network.layers.2.weight_g"), st("geometry.network.layers.2.weight_v")), geo_b2=st("geometry.network.layers.2.bias"),
        rad_center=center, rad_scale=scale, rad_params=t_rad, rad_mask=np.ones(32, np.float32), rad_sh_mask=st("radiance.sh_mask")[0],
        rad_W=[st(f"radiance.network.layers.{i}.weight") for i in (0, 2, 4)], rad_b=[st(f"radiance.network.layers.{i}.bias") for i in (0, 2, 4)],
        mat_W=Ws, mat_b=bs, env_base=G["hdri"])


def gpu_scene(G, tag, dev="cuda:0"):
    """(RenderStep, material, emitter, rays) of the golden scene on the HIP path: the reference's state_dict loads into the
    package's modules under the reference's own keys (checkpoint.py)."""
    import torch
    from intrinsicavatar_amd import fields, pbr, render, checkpoint
    from intrinsicavatar_amd.deformer import SNARFDeformer
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    dfm = SNARFDeformer(T(G["rig_lbs_voxel_final"]), T(G["rig_offset_kernel"]), T(G["rig_scale_kernel"]), T(G["rig_cano_bbox"]))
    dfm.prepare(T(G["rig_tfs"]), T(G["rig_w2s"]))
    geo, rad = fields.VolumeSDF(seed=0).to(dev), fields.VolumeRefDirRadiance(seed=1).to(dev)
    dens, mat = fields.LaplaceDensity(beta_init=0.3).to(dev), fields.VolumeMaterial(seed=2).to(dev)
    t_geo, t_rad = hash_tables()
    sd = {"model." + str(k): torch.from_numpy(np.ascontiguousarray(G["state_" + str(k)])) for k in G["state_keys"] if "state_" + str(k) in G.files}
    sd["model.geometry.encoding.encoding.encoding.params"] = torch.from_numpy(t_geo)
    sd["model.radiance.xyz_encoding.encoding.encoding.params"] = torch.from_numpy(t_rad)
    rs = render.RenderStep(geo, rad, dens, dfm, T(G[tag + "_occ_binaries"]), T(G[tag + "_occ_aabb"]), float(G["render_step_size"]))
    checkpoint.load_reference_state_dict(rs, sd, material=mat, strict=False)
    for m in (geo, rad):
        m.update_step(250, 25000)                                  # systems/base.py:150
        m.prepare_bbox(dfm.bbox)
    env = pbr.EnvironmentLightTensor(T(G["hdri"]))
    env.update_pdf()
    return rs, mat, env, T(G["rays"])


# ----------------------------------------------------------------------------- golden_backward.npz: summaries of a table gradient
SUBSAMPLE = 32


def _S():
    from intrinsicavatar_amd import synthetic as S
    return S


def subsample_mask(idx):
    """1-in-SUBSAMPLE selection of table entries by a hash of the FLOAT index (numpy uint64 arithmetic, closed form)."""
    h = (idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
    return (h % np.uint64(SUBSAMPLE)) == 0


def level_offsets():
    from tests import torch_ref as TR
    return np.asarray(TR.hash_cfg()[0], dtype=np.int64)


def table_gradient_summary(g):
    """g: flat float gradient of one table's `params` (entries x 2 features) -> dict of arrays (see the module docstring)."""
    off = level_offsets() * 2                                        # float offsets of the 16 levels (+ end)
    g64 = g.astype(np.float64)
    probe = _S().hash_table_values(g.size, 4242, 1.0).astype(np.float64)
    out = dict(level_sum=np.array([g64[off[i]:off[i + 1]].sum() for i in range(len(off) - 1)]),
               level_l1=np.array([np.abs(g64[off[i]:off[i + 1]]).sum() for i in range(len(off) - 1)]),
               level_l2=np.array([np.sqrt((g64[off[i]:off[i + 1]] ** 2).sum()) for i in range(len(off) - 1)]),
               level_probe=np.array([(g64[off[i]:off[i + 1]] * probe[off[i]:off[i + 1]]).sum() for i in range(len(off) - 1)]),
               level_nnz=np.array([int((g[off[i]:off[i + 1]] != 0).sum()) for i in range(len(off) - 1)]))
    nz = np.nonzero(g)[0]
    keep = nz[subsample_mask(nz)]
    out["sub_index"] = keep.astype(np.int32)
    out["sub_value"] = g[keep].astype(np.float32)
    return out


# ----------------------------------------------------------------------------- outliers of the SDF gradient, explained sample by sample
def explain_gradient_outliers(rs, pts_gpu, pts_ref, grad_ref, thresh, max_shift=1e-3, k_ulp=16, dev="cuda:0"):
    """Why a few samples' SDF gradients differ by O(0.1) between the HIP path and a CPU implementation of the same step
    (tools/grad_outlier_probe.py, profiles/r05_grad_outliers.json: 241 of 211 195 samples above 10 x p99, every one of them with
    the HIP field AT THE ORACLE'S sample point within 5e-7 of the oracle's gradient):
      A. the two pipelines do not evaluate the field at bit-identical sample points -- the K2 / K4 edges come from inverting a CDF of
         fp32 weights, 6 % of the samples differ in t by an ulp or more -- and the gradient of this field is steep (Softplus beta = 100)
         and piecewise constant in the cell index of every hash level, so a shift of 1e-6 can change it by 0.1;
      B. at an IDENTICAL point two implementations can still pick different cells when a level coordinate x01 * scale + 0.5 lies within
         an ulp of an integer (fmaf here, multiply-then-add in torch).
    Checked per outlier, for posed-space sample points `pts_gpu` (this path's), `pts_ref` (the other implementation's) [m,3] and the
    other implementation's gradient `grad_ref` [m,3]:
      same_point    |pts_gpu - pts_ref|_inf <= max_shift (it IS the same sample);
      field_agrees  A: the HIP field evaluated AT THE OTHER IMPLEMENTATION'S point returns its gradient within `thresh`;
      face_flip     B: otherwise -- a level coordinate of that point's root is within k_ulp ulp of a cell face, and evaluated on the
                    other side of that face (64 ulp across: 1e-5 of a cell) the HIP field returns the other gradient within `thresh`.
      near_tie      C: otherwise -- the point has a second candidate root whose SDF is within 2e-5 of the selected minimum (the min over
                    the candidates, snarf_deformer.py:192-231, is decided by the last bits) and THAT candidate's gradient is the other one.
      jump_nearby   D: otherwise -- within 4e-6 m of that point (probes along the axes and the diagonal) the HIP path returns the other
                    gradient: the sample sits on a discontinuity of the composed map point -> root -> gradient that the hash-face test
                    does not see, e.g. where Broyden's |g| < 1e-5 stop (fuse_cuda_kernel_fast.cu:252-452) takes one more iteration and
                    the root moves by ~1e-5 (seen on 1 of 12 712 samples of the golden frame).
    explained = same_point and (A or B or C or D).  -> (explained [m] bool, dict of the conditions + residuals)."""
    import torch
    from tests import torch_ref as TR
    geo, dfm = rs.geometry, rs.deformer
    pg = torch.as_tensor(pts_gpu, device=dev).float().contiguous()
    pr = torch.as_tensor(pts_ref, device=dev).float().contiguous()
    ref = torch.as_tensor(grad_ref, device=dev).float()
    m = pg.shape[0]
    if m == 0:
        z = np.zeros(0, bool)
        return z, dict(same_point=z, field_agrees=z, face_flip=z, near_tie=z, jump_nearby=z, cell_changes=z, residual=np.zeros(0), shift=np.zeros(0))
    with torch.no_grad():
        dg = dfm.deform(pg, geo, with_grad=True, with_feature=False)
        dr = dfm.deform(pr, geo, with_grad=True, with_feature=False, want_fwd=True)
        shift = (pg - pr).abs().max(-1)[0]
        residual = (dr["sdf_grad"] - ref).abs().max(-1)[0]
        scales = torch.tensor(TR.hash_cfg()[2], device=dev, dtype=torch.float32)          # [16]

        def level_pos(xc):
            x01 = ((xc - geo.center) / geo.scale + 0.5).float()
            return x01, x01[:, None, :] * scales[None, :, None] + 0.5                     # [m,3], [m,16,3]
        _, pos_g = level_pos(dg["pts_cano"])
        x01, pos = level_pos(dr["pts_cano"])
        cell_changes = (torch.floor(pos_g) != torch.floor(pos)).reshape(m, -1).any(1)
        same_point = shift <= max_shift
        field_agrees = residual <= thresh
        # B: nudge across a near face, at the other implementation's point
        sel = dr["sel"].long().clamp(min=0)
        c2w = dr["fwd_J"].reshape(-1, 3, 3)[dr["cand_src"].long()[sel]]
        face = torch.round(pos)
        ulp = torch.nextafter(pos.abs(), torch.full_like(pos, float("inf"))) - pos.abs()
        near = (pos - face).abs() <= k_ulp * ulp
        best = torch.full((m,), float("inf"), device=dev)
        lv, ax = torch.nonzero((near & ~field_agrees[:, None, None]).any(0), as_tuple=True)
        for l_, a_ in zip(lv.tolist(), ax.tolist()):
            who = near[:, l_, a_] & ~field_agrees
            side = torch.where(pos[:, l_, a_] >= face[:, l_, a_], -1.0, 1.0)              # to the OTHER side of the face
            target = face[:, l_, a_] + side * 64 * ulp[:, l_, a_]
            xt = x01.clone()
            xt[:, a_] = torch.where(who, (target - 0.5) / scales[l_], x01[:, a_])
            _, g, _ = geo(((xt - 0.5) * geo.scale + geo.center).contiguous(), with_grad=True, with_feature=True)
            err = ((c2w * g[:, None, :]).sum(-1) - ref).abs().max(-1)[0]
            best = torch.where(who, torch.minimum(best, err), best)
        face_flip = ~field_agrees & (best <= thresh)
        # C: another candidate within 2e-5 of the minimum SDF carries the other gradient
        near_tie = torch.zeros(m, dtype=torch.bool, device=dev)
        left = torch.nonzero(~field_agrees & ~face_flip)[:, 0]
        if left.numel() > 0:
            pl = pr[left].contiguous()
            cand_x, cand_src, cnt, start, Q, fwd, _ = dfm._candidates(pl, with_src=True, want_fwd=True)
            _, cg, cf = geo(cand_x, with_grad=True, with_feature=True)
            cw = (fwd.reshape(-1, 3, 3)[cand_src.long()] * cg[:, None, :]).sum(-1)          # pushed-forward gradient of every candidate
            csdf = cf[:, 0]
            for j in range(left.numel()):
                a, n_ = int(start[j]), int(cnt[j])
                if n_ < 2:
                    continue
                sd, gw = csdf[a:a + n_], cw[a:a + n_]
                close = (sd - sd.min()) <= 2e-5
                hit = close & ((gw - ref[left[j]][None]).abs().max(-1)[0] <= thresh)
                near_tie[left[j]] = bool(close.sum() >= 2 and hit.any())
        jump_nearby = torch.zeros(m, dtype=torch.bool, device=dev)
        left = torch.nonzero(~field_agrees & ~face_flip & ~near_tie)[:, 0]
        if left.numel() > 0:
            dirs = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [0.57735, 0.57735, 0.57735]], device=dev)
            for step in (1e-6, 2e-6, 4e-6):
                for sgn in (-1.0, 1.0):
                    for dv in dirs:
                        q = (pr[left] + sgn * step * dv[None]).contiguous()
                        e_ = (dfm.deform(q, geo, with_grad=True, with_feature=False)["sdf_grad"] - ref[left]).abs().max(-1)[0]
                        jump_nearby[left] |= e_ <= thresh
        explained = same_point & (field_agrees | face_flip | near_tie | jump_nearby)
    c = lambda t: t.cpu().numpy()      # noqa: E731
    return c(explained), dict(same_point=c(same_point), field_agrees=c(field_agrees), face_flip=c(face_flip), near_tie=c(near_tie),
                              jump_nearby=c(jump_nearby), cell_changes=c(cell_changes), residual=c(torch.where(field_agrees, residual, torch.minimum(residual, best))),
                              shift=c(shift))


def assert_normal_outliers_explained(rs, rays, out, ref, dev="cuda:0"):
    """For a RenderStep.forward result `out` and the oracle's render_step result `ref` WITH IDENTICAL SAMPLE SETS: every sample whose SDF
    gradient is more than 10 x p99 away from the oracle's is explained by explain_gradient_outliers (position / cell-face flip / near tie /
    jump within 4e-6 m), and the outlier pixels of comp_normal (10 x p99) are the rays of those samples.  -> number of outlier samples."""
    import torch
    g_gpu, g_ref = out["sdf_grad"].cpu().numpy(), ref["sdf_grad"]
    both = out["valid"].cpu().numpy()
    e = np.where(both, np.abs(g_gpu - g_ref).max(-1), 0.0)
    thresh = 10.0 * float(np.quantile(e, 0.99))
    idx = np.nonzero(e > thresh)[0]
    r_smpl = rs.deformer.transform_rays_w2s(rays.float())
    ri = out["ray_indices"].long()
    sel_ = torch.from_numpy(idx).to(ri.device)
    mid_g = (out["t_starts"] + out["t_ends"]) / 2.0
    mid_r = (torch.from_numpy(ref["t_starts"]).to(dev) + torch.from_numpy(ref["t_ends"]).to(dev)) / 2.0
    pts_g = (r_smpl[ri, :3] + r_smpl[ri, 3:6] * mid_g[:, None])[sel_]
    pts_r = (r_smpl[ri, :3] + r_smpl[ri, 3:6] * mid_r[:, None])[sel_]
    explained, why = explain_gradient_outliers(rs, pts_g, pts_r, g_ref[idx], thresh)
    assert explained.all(), (idx[~explained].tolist(), {k: v[~explained].tolist() for k, v in why.items()}, thresh)
    flip_rays = set(ri.cpu().numpy()[idx].tolist())
    en = np.abs(out["comp_normal"].cpu().numpy() - ref["comp_normal"]).max(-1)
    bad_px = set(np.nonzero(en > 10.0 * float(np.quantile(en, 0.99)))[0].tolist())
    assert bad_px <= flip_rays, sorted(bad_px - flip_rays)
    return int(idx.size)
