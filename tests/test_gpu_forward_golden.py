"""GPU (MI355X): the HIP path against the reference's OWN IntrinsicAvatarModel.forward_ (models/intrinsic_avatar.py:950-1651),
run on CPU in the build container by tests/golden/make_golden_forward.py (tests/golden/golden_forward.npz): same scene (the
reference's state_dict loads under its own keys), same rays, same explicit random tensors; render modes light / uniform_light /
mis / mats in the eval form and `light` in train() mode; plus the per-frame pieces the model runs before forward_
(ForwardDeformer.switch_to_explicit + precompute, prepare_test_occupancy_grid :307-381).

Bars as in tests/test_forward_golden_cpu.py (which holds the CPU oracle to the same fixture): output keys / shapes / dtypes of
the reference's dict; sample counts within 0.5 %; step-5 maps 2e-3 (normals 4e-3) on >= 98.5 % of the pixels; Monte-Carlo
images 2 % + 2e-2 on >= 97 % of the pixels and 2 % in the mean."""
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


def _check_common(d, ref, mode):
    for k in ref:
        assert tuple(d[k].shape) == ref[k].shape, (k, tuple(d[k].shape), ref[k].shape)
        assert N(d[k]).dtype.kind == ref[k].dtype.kind, (k, N(d[k]).dtype, ref[k].dtype)
    n_ref = int(ref["num_samples"][0])
    assert abs(int(d["num_samples"][0]) - n_ref) <= 0.005 * n_ref, (int(d["num_samples"][0]), n_ref)
    assert (N(d["rays_valid"]) == ref["rays_valid"]).mean() >= 0.995
    for k in ("comp_rgb_bg", "comp_albedo_bg", "comp_metallic_bg", "comp_roughness_bg", "rays_valid_bg", "num_samples_bg"):
        np.testing.assert_allclose(N(d[k]).astype(np.float64), ref[k].astype(np.float64), atol=1e-6, err_msg=k)
    for k, tol in (("comp_rgb", 2e-3), ("comp_normal", 4e-3), ("comp_albedo", 2e-3), ("comp_roughness", 2e-3), ("comp_metallic", 2e-3),
                   ("opacity", 2e-3), ("comp_rgb_full", 4e-3), ("comp_albedo_full", 2e-3)):
        _close(k, N(d[k]), ref[k], tol, mean_tol=5e-4)
    _close("depth", N(d["depth"]), ref["depth"], 5e-3)


def _check_mc_images(d, ref, keys=("comp_rgb_phys", "comp_demod_phys"), rows=None):
    hit = ref["rays_valid"][:, 0]
    for k in keys:
        a, b = N(d[k]), ref[k]
        if rows is not None:
            a, b, h = a[:rows], b[:rows], hit[:rows]
        else:
            h = hit
        tol = 2e-2 * np.abs(b).max(-1) + 2e-2
        err = np.abs(a - b).max(-1)
        assert (err <= tol).mean() >= 0.97, (k, float((err > tol).mean()), float(err.max()))
        assert abs(a[h].mean() - b[h].mean()) <= 2e-2 * abs(b[h].mean()), (k, a[h].mean(), b[h].mean())


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
    _check_common(d, ref, mode)
    _check_mc_images(d, ref)
    if mode == "uniform_light":
        _close("visibility", N(d["visibility"]), ref["visibility"], 3e-2, frac=0.97)


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
    assert abs(n_gpu - n_ref) <= 0.005 * n_ref
    per_sample = ("sdf_samples", "sdf_grad_samples", "sdf_laplace_samples", "weights", "points", "intervals", "ray_indices")
    _check_common({k: v for k, v in d.items() if k not in per_sample}, {k: v for k, v in ref.items() if k not in per_sample}, "light")
    for k in ("normals_orientation_loss_map", "albedo_smoothness_loss_map", "roughness_smoothness_loss_map", "metallic_smoothness_loss_map"):
        a, b = N(d[k]), ref[k]
        assert a.shape == b.shape
        if n_gpu == n_ref:            # the jitter noise is indexed by sample: only comparable when the sample sets coincide
            _close(k, a, b, 5e-3 * max(1.0, float(np.abs(b).max())), frac=0.97)
    if n_gpu == n_ref and np.array_equal(N(d["ray_indices"]), ref["ray_indices"]):
        # K2 inserts its edges by inverting a CDF of fp32 weights: a last-bit difference in a weight moves an edge by an ulp or two
        assert (np.abs(N(d["points"]) - ref["points"]) <= 2e-5).mean() >= 0.999
        assert (np.abs(N(d["intervals"]) - ref["intervals"]) <= 2e-5).mean() >= 0.999 and np.abs(N(d["intervals"]) - ref["intervals"]).max() < 1e-3
        _close("weights", N(d["weights"])[:, None], ref["weights"][:, None], 2e-3, frac=0.99)
        _close("sdf_samples", N(d["sdf_samples"])[:, None], ref["sdf_samples"][:, None], 1e-4, frac=0.99)
        _close("sdf_grad_samples", N(d["sdf_grad_samples"]), ref["sdf_grad_samples"], 5e-3, frac=0.99)
        assert float(np.abs(ref["sdf_laplace_samples"]).max()) == 0.0 and float(d["sdf_laplace_samples"].abs().max()) == 0.0
    # per-point light: image on the rays before the first pixel that disagrees (a fg / bg flip re-pairs every later uniform)
    a, b = N(d["comp_rgb_phys"]), ref["comp_rgb_phys"]
    tol = 2e-2 * np.abs(b).max(-1) + 2e-2
    bad = np.nonzero(np.abs(a - b).max(-1) > tol)[0]
    first_bad = int(bad[0]) if bad.size else a.shape[0]
    assert first_bad >= 0.3 * a.shape[0], first_bad


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
