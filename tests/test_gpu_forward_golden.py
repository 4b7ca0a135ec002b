"""GPU (MI355X): the HIP path against the reference's OWN IntrinsicAvatarModel.forward_ (models/intrinsic_avatar.py:950-1651),
run on CPU in the build container by tests/golden/make_golden_forward.py (tests/golden/golden_forward.npz): same scene (the
reference's state_dict loads under its own keys), same rays, same explicit random tensors; render modes light / uniform_light /
mis / mats in the eval form and `light` in train() mode; plus the per-frame pieces the model runs before forward_
(ForwardDeformer.switch_to_explicit + precompute, prepare_test_occupancy_grid :307-381).

Bars (round 4): every float key of the output dict is held to (max, p99, mean) of its per-pixel absolute difference, set at 3 x what
tools/parity_table.py observed on the MI355X (profiles/r04_parity_table.json; BASELINE.md section 3 quotes the table) -- the maximum
is a HARD cap over all 784 pixels, not a fraction-of-pixels catch-all; discrete outputs (sample counts, rays_valid*, ray indices)
are compared exactly, with a stated upper bound on flips (observed: none).  For scale: the CPU oracle differs from the same fixture by
1.3e-4 (comp_rgb) / 8.5e-4 (comp_normal) / 1.3e-3 (comp_rgb_phys)."""
import numpy as np
import pytest
import torch

from tests import forward_golden as FG

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def G():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from intrinsicavatar_amd import build
    build.build()
    return FG.load()


def _close(name, a, b, tol, frac=0.985, mean_tol=None):
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    err = err.reshape(err.shape[0], -1).max(-1)
    ok = err <= tol
    assert ok.mean() >= frac, (name, float(ok.mean()), float(err.max()))
    if mean_tol is not None:
        assert err.mean() < mean_tol, (name, float(err.mean()))


# (max, p99, mean) of the per-pixel absolute difference to the reference's forward_: 3 x the MI355X observation, worst eval run
BARS_EVAL = {
    "comp_rgb": (2.8e-3, 9.5e-5, 1.2e-5), "comp_rgb_full": (2.0e-3, 6.5e-5, 8.4e-6), "comp_normal": (1.3e-2, 3.6e-4, 5.3e-5),
    "comp_albedo": (1.8e-6, 1.1e-6, 1.3e-7), "comp_albedo_full": (1.8e-6, 1.1e-6, 1.3e-7),
    "comp_roughness": (2.2e-6, 1.2e-6, 1.3e-7), "comp_roughness_full": (7.2e-7, 4.7e-7, 1.2e-7),
    "comp_metallic": (2.2e-6, 1.4e-6, 1.3e-7), "comp_metallic_full": (7.2e-7, 5.4e-7, 1.2e-7),
    "opacity": (4.2e-6, 2.7e-6, 2.4e-7), "depth": (7.2e-6, 5.8e-6, 6.9e-7),
}
BARS_TRAIN = {       # train(): jittered near plane + the training occupancy grid -- other samples than the eval runs
    "comp_rgb": (6.5e-3, 9.8e-5, 2.0e-5), "comp_rgb_full": (4.3e-3, 7.0e-5, 1.5e-5), "comp_normal": (3.3e-2, 3.1e-4, 9.0e-5),
    "comp_albedo": (2.0e-6, 9.3e-7, 1.3e-7), "comp_albedo_full": (2.0e-6, 9.3e-7, 1.3e-7),
    "comp_roughness": (2.6e-6, 1.2e-6, 1.3e-7), "comp_roughness_full": (7.2e-7, 3.8e-7, 1.2e-7),
    "comp_metallic": (2.6e-6, 1.2e-6, 1.3e-7), "comp_metallic_full": (7.2e-7, 5.4e-7, 1.3e-7),
    "opacity": (4.9e-6, 2.2e-6, 2.5e-7), "depth": (1.1e-5, 5.8e-6, 7.5e-7),
}
# Monte-Carlo images, per run (a visibility sample on the other side of a threshold moves a pixel by Lo / spp).  The MAXIMA of these maps are
# set by single flipped samples and therefore move when any fp32 operation order upstream changes (round 6: the effective weights of the
# heads come out of ia_effective_weights instead of a torch chain -- last-bit differences -- and one more sample flipped in light_64_gi /
# uniform_light_512_gi: comp_demod_phys 1.7e-3 -> 5.7e-3, comp_demod_phys_full 8e-4 -> 3.8e-3 at unchanged p99 / mean); the p99 and mean
# columns are the stable part of these bars
BARS_MC = {
    "light_16_nogi": {"comp_rgb_phys": (4.0e-3, 1.1e-4, 9.6e-6), "comp_demod_phys": (7.7e-3, 3.7e-4, 2.6e-5), "comp_rgb_phys_full": (4.0e-3, 1.3e-4, 1.2e-5), "comp_demod_phys_full": (3.8e-3, 3.1e-4, 1.8e-5)},
    "light_64_gi": {"comp_rgb_phys": (2.5e-3, 1.6e-4, 1.2e-5), "comp_demod_phys": (1.7e-2, 5.3e-4, 5.9e-5), "comp_rgb_phys_full": (1.4e-3, 1.4e-4, 9.6e-6), "comp_demod_phys_full": (6.2e-4, 1.8e-4, 9.1e-6)},
    "uniform_light_512_gi": {"comp_rgb_phys": (2.3e-2, 2.8e-4, 4.2e-5), "comp_demod_phys": (7.0e-2, 7.6e-4, 1.4e-4), "comp_rgb_phys_full": (1.5e-2, 2.5e-4, 2.9e-5), "comp_demod_phys_full": (1.15e-2, 1.8e-4, 2.5e-5)},
    "mis_16_gi": {"comp_rgb_phys": (1.2e-2, 7.2e-4, 4.8e-5), "comp_demod_phys": (3.5e-2, 3.2e-3, 1.8e-4), "comp_rgb_phys_full": (2.0e-2, 1.3e-3, 6.4e-5), "comp_demod_phys_full": (4.4e-2, 1.9e-3, 1.4e-4)},
    "mats_16_gi": {"comp_rgb_phys": (2.2e-2, 1.3e-3, 9.4e-5), "comp_demod_phys": (6.7e-2, 5.1e-3, 3.2e-4), "comp_rgb_phys_full": (3.1e-2, 2.0e-3, 1.2e-4), "comp_demod_phys_full": (7.6e-2, 2.4e-3, 2.3e-4)},
    "light_16_gi_train": {"comp_rgb_phys": (4.7e-4, 1.3e-4, 7.2e-6), "comp_demod_phys": (1.8e-3, 3.5e-4, 2.3e-5), "comp_rgb_phys_full": (6.0e-4, 1.4e-4, 8.7e-6), "comp_demod_phys_full": (1.1e-3, 1.9e-4, 1.3e-5)},
}


def _held(name, a, b, bar):
    """per-pixel absolute difference within (max, p99, mean); the max is a hard cap over every pixel."""
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    err = err.reshape(err.shape[0], -1).max(-1)
    got = (float(err.max()), float(np.quantile(err, 0.99)), float(err.mean()))
    assert got[0] <= bar[0] and got[1] <= bar[1] and got[2] <= bar[2], (name, got, bar)


def _check_common(d, ref, tag, bars=None):
    for k in ref:
        assert tuple(d[k].shape) == ref[k].shape, (k, tuple(d[k].shape), ref[k].shape)
        assert N(d[k]).dtype.kind == ref[k].dtype.kind, (k, N(d[k]).dtype, ref[k].dtype)
    # discrete outputs: exact on the MI355X; upper bounds on what a sample at a threshold may flip
    assert abs(int(d["num_samples"][0]) - int(ref["num_samples"][0])) <= 2, (int(d["num_samples"][0]), int(ref["num_samples"][0]))
    for k in ("rays_valid", "rays_valid_phys", "rays_valid_full", "rays_valid_phys_full"):
        assert int((N(d[k]) != ref[k]).sum()) <= 1, k
    for k in ("rays_valid_bg", "rays_valid_phys_bg", "num_samples_bg"):
        assert np.array_equal(N(d[k]), ref[k]), k
    for k in ("comp_rgb_bg", "comp_albedo_bg", "comp_metallic_bg", "comp_roughness_bg"):
        np.testing.assert_allclose(N(d[k]).astype(np.float64), ref[k].astype(np.float64), atol=1e-7, err_msg=k)
    for k, bar in (bars or BARS_EVAL).items():
        _held(k, N(d[k]), ref[k], bar)
    for k, bar in BARS_MC[tag].items():
        _held(k, N(d[k]), ref[k], bar)
    # the Monte-Carlo images also agree in the mean over the hit pixels
    hit = ref["rays_valid"][:, 0]
    for k in ("comp_rgb_phys", "comp_demod_phys"):
        a, b = N(d[k]), ref[k]
        assert abs(a[hit].mean() - b[hit].mean()) <= 2e-3 * abs(b[hit].mean()), (k, a[hit].mean(), b[hit].mean())


@pytest.mark.parametrize("tag", list(FG.RUNS))
def test_forward_eval_vs_the_references_own_forward(G, tag):
    mode, spp, gi = FG.RUNS[tag]
    rs, mat, env, rays = FG.gpu_scene(G, tag)
    rnd = FG.explicit_randoms(G, tag)
    light_u = rnd["stratified_u"] if mode == "uniform_light" else rnd["light_u"]
    scatter_u = None
    if "scatter_u" in rnd:            # the foreground count may differ by a few re-samples: spare uniforms at the end
        g = torch.Generator().manual_seed(0)
        scatter_u = torch.cat([torch.from_numpy(rnd["scatter_u"]), torch.rand((4096, 6), generator=g)]).to(DEV)
    d = rs.forward_(rays, mat, env, spp, T(light_u), T(rnd["shuffle_u"]) if "shuffle_u" in rnd else None,
                    background_color=T(G["background_color"]), global_illumination=gi, render_mode=mode, scatter_u=scatter_u)
    ref = {str(k): G[f"{tag}_out_{k}"] for k in G[tag + "_out_keys"]}
    assert sorted(d) == sorted(ref), sorted(set(d) ^ set(ref))
    _check_common(d, ref, tag)
    if mode == "uniform_light":
        _held("visibility", N(d["visibility"]), ref["visibility"], (4.6e-3, 1.3e-5, 9.7e-6))


def test_forward_train_mode_vs_the_references_own_forward(G):
    """train(): stratified near plane, the training occupancy grid, the material jitter pass and the four loss maps, per-point
    light directions, and the per-sample keys of the training dict (:1519-1597)."""
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
    assert sorted(d) == sorted(ref), sorted(set(d) ^ set(ref))
    n_ref = int(ref["num_samples"][0])
    n_gpu = int(d["num_samples"][0])
    per_sample = ("sdf_samples", "sdf_grad_samples", "sdf_laplace_samples", "weights", "points", "intervals", "ray_indices")
    _check_common({k: v for k, v in d.items() if k not in per_sample}, {k: v for k, v in ref.items() if k not in per_sample}, tag,
                  bars=BARS_TRAIN)
    # the sample SET is the reference's (observed on the MI355X: identical); the per-sample keys are only comparable then
    assert n_gpu == n_ref and np.array_equal(N(d["ray_indices"]), ref["ray_indices"])
    for k, bar in (("normals_orientation_loss_map", (1.6e-2, 1.3e-4, 4.1e-5)), ("albedo_smoothness_loss_map", (1.3e-10, 4.4e-11, 5e-12)),
                   ("roughness_smoothness_loss_map", (8.8e-11, 4.7e-11, 3.6e-12)), ("metallic_smoothness_loss_map", (6.4e-11, 3.1e-11, 2.9e-12)),
                   # K2 inserts its edges by inverting a CDF of fp32 weights: a last-bit difference in a weight moves an edge by an ulp or two
                   ("points", (4.6e-5, 2.9e-6, 1.8e-7)), ("intervals", (9.2e-5, 4.3e-6, 3.1e-7)), ("weights", (2.4e-5, 8.6e-6, 4.5e-7)),
                   ("sdf_samples", (5.2e-5, 1.7e-6, 1.9e-7)),
                   # the SDF gradient jumps across the faces of the hash grid's cells: a sample within an ulp of a face may take the other side
                   ("sdf_grad_samples", (2.2, 3.0e-3, 4.0e-4))):
        a, b = N(d[k]), ref[k]
        _held(k, a if a.ndim > 1 else a[:, None], b if b.ndim > 1 else b[:, None], bar)
    assert float(np.abs(ref["sdf_laplace_samples"]).max()) == 0.0 and float(d["sdf_laplace_samples"].abs().max()) == 0.0
    # ---- the outliers of the SDF gradient are hash-cell-face flips, and nothing else (checked, not asserted by comment;
    # tests/forward_golden.explain_gradient_outliers): for every sample whose gradient is more than 10 x p99 away from the reference's,
    # it is the same sample (|shift| <= 1e-3), and the HIP field AT THE REFERENCE'S point returns the reference's gradient -- the difference is
    # the sample position (K2's CDF inversion of fp32 weights) -- or, at an identical point, the other side of a hash-cell face a few ulp away does
    g_gpu, g_ref = N(d["sdf_grad_samples"]), ref["sdf_grad_samples"]
    err = np.abs(g_gpu - g_ref).max(-1)
    thresh = 10.0 * float(np.quantile(err, 0.99))
    out_s = np.nonzero(err > thresh)[0]
    r_smpl = rs.deformer.transform_rays_w2s(rays.float())
    ri = d["ray_indices"].long()
    sel_ = torch.from_numpy(out_s).to(DEV)
    pts_gpu = (r_smpl[ri, :3] + r_smpl[ri, 3:6] * d["points"][:, None])[sel_]
    pts_ref = (r_smpl[ri, :3] + r_smpl[ri, 3:6] * T(ref["points"])[:, None])[sel_]
    explained, why = FG.explain_gradient_outliers(rs, pts_gpu, pts_ref, g_ref[out_s], thresh)
    assert explained.all(), ("sdf_grad_samples outliers that are NOT cell-face flips", out_s[~explained].tolist(),
                             {k: v[~explained].tolist() for k, v in why.items()}, thresh)
    # the pixel maps that carry the gradient: an outlier pixel (10 x p99) owns an outlier sample
    rays_with_flip = set(N(ri)[out_s].tolist())
    for k in ("comp_normal", "normals_orientation_loss_map"):
        e = np.abs(N(d[k]) - ref[k]).reshape(ref[k].shape[0], -1).max(-1)
        bad_px = np.nonzero(e > 10.0 * float(np.quantile(e, 0.99)))[0]
        assert set(bad_px.tolist()) <= rays_with_flip, (k, sorted(set(bad_px.tolist()) - rays_with_flip))
    print(f"cell-face check: {out_s.size} gradient outliers of {err.size} samples (> {thresh:.2e}), all explained: "
          f"{int(why['field_agrees'].sum())} by position, {int(why['face_flip'].sum())} cell-face flips, {int(why['near_tie'].sum())} near ties, "
          f"{int(why['jump_nearby'].sum())} jumps within 4e-6 m")


def test_per_frame_preparation_vs_the_reference(G):
    """what the model does per frame before forward_: ForwardDeformer.precompute on the reference's own offset / scale kernels
    (voxel_J, K10) and prepare_test_occupancy_grid (:307-381: 3 jittered points per voxel -> alpha -> max -> 3^3 max-pool ->
    threshold clamp(mean, max = grid_prune_occ_thre) -> largest connected component)."""
    from intrinsicavatar_amd import occ_grid, render
    tag = "light_16_nogi"
    rs, mat, env, rays = FG.gpu_scene(G, tag)
    vj = N(rs.deformer.voxel_J_cl)[0]                                    # [D,H,W,12]
    np.testing.assert_allclose(np.moveaxis(vj, -1, 0), G["rig_ref_voxel_J"][0], rtol=0, atol=1e-6)
    rnd = FG.explicit_randoms(G, tag)
    aabb = T(G[tag + "_occ_aabb"][0])
    v = G["rig_vertices"][0]                                             # get_bbox_from_smpl (snarf_deformer.py:24-35)
    c, s = (v.max(0) + v.min(0)) / 2, ((v.max(0) - v.min(0)) / 2).max() * 1.2
    np.testing.assert_allclose(np.concatenate([c - s, c + s]), G[tag + "_occ_aabb"][0], atol=1e-6)
    beta = rs.density.get_beta().detach().reshape(1)

    def occ_eval_fn(x):
        return render.laplace_alpha(rs.deformer.deform(x, rs.geometry)["sdf"], rs.render_step_size, beta)
    _, binaries = occ_grid.compute_test_occupancy_grid(occ_eval_fn, aabb, 64, 3, 0.001, T(rnd["occ_jitter"].reshape(-1, 3, 3)))
    ref = G[tag + "_occ_binaries"]
    agree = (N(binaries) == ref).mean()
    assert agree >= 0.9995 and abs(int(binaries.sum()) - int(ref.sum())) <= 0.01 * ref.sum(), (agree, int(binaries.sum()), int(ref.sum()))
