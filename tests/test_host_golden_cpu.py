"""CPU: the oracle's restatements of the reference's HOST-side logic against vectors produced by the reference's own
Python (tests/golden/golden_host.npz <- tests/golden/make_golden_host.py, which imports models/pbr/utils.py,
models/occ_grid/temporal_occ_grid.py, models/utils.py, models/rf/density.py from /root/reference in the build container)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(f"{golden_dir}/golden_host.npz")


@pytest.mark.parametrize("spp", [8, 64])
def test_oracle_sample_volume_interaction_vs_reference(oracle, G, spp):
    """oracle/render_ref.py sample_volume_interaction == models/pbr/utils.py:70-229 run over the reference's own K1."""
    from oracle import render_ref as R
    p = f"svi{spp}_"
    extras = {k: G[p + "in_" + k] for k in ("weights", "sdf", "alphas", "normals", "albedo", "roughness", "metallic")}
    rpi, rri, rw, fg, bg, ex, _ = R.sample_volume_interaction(G[p + "rays_o"], G[p + "rays_d"], G[p + "ray_indices"], G[p + "t_starts"],
                                                             G[p + "t_ends"], int(G[p + "n_rays"]), spp, G[p + "transmittance"], extras)
    np.testing.assert_array_equal(rpi, G[p + "resampled_packed_info"])
    np.testing.assert_array_equal(rri, G[p + "resampled_ray_indices"])
    np.testing.assert_array_equal(fg, G[p + "fg_indices"])
    np.testing.assert_array_equal(bg, G[p + "bg_indices"])
    np.testing.assert_array_equal(rw, G[p + "resampled_weights"])
    assert sorted(ex.keys()) == list(G[p + "extras_keys"])
    for k in ex:
        np.testing.assert_array_equal(ex[k], G[p + "out_" + k], err_msg=k)
    assert len(fg) > 100 and len(bg) > 100


def test_oracle_occupancy_update_vs_reference(G):
    """oracle/occgrid_ref.py (EMA max, 3^3 max-pool, threshold, largest connected component) ==
    TemporalOccGridEstimator._update + max_connected_component of the reference, two consecutive updates of level 1."""
    from oracle import occgrid_ref as R
    res = int(G["occ_res"])
    cells = res ** 3
    occ1 = G["occ_eval1"].astype(np.float32)
    occs = np.maximum(np.zeros(cells, np.float32) * np.float32(0.8), occ1)
    np.testing.assert_array_equal(occs, G["occ_occs_after1"][cells:])
    b1, _ = R.binarize(occs, (res, res, res), 0.001, True)
    np.testing.assert_array_equal(b1, G["occ_binaries_after1"][1])
    assert not G["occ_binaries_after1"][0].any()
    occs2 = np.maximum(occs * np.float32(0.8), occ1 * np.float32(0.5))
    np.testing.assert_array_equal(occs2, G["occ_occs_after2"][cells:])
    b2, _ = R.binarize(occs2, (res, res, res), 0.001, True)
    np.testing.assert_array_equal(b2, G["occ_binaries_after2"][1])
    # the filter removed something: the second blob is occupied before the component filter, not after
    b_nofilter, _ = R.binarize(occs, (res, res, res), 0.001, False)
    assert b_nofilter.sum() > b1.sum() > 100
    # max_connected_component on its own: label volume after the reference's sweeps
    comp = R.connected_component_labels(G["mcc_in"][0])
    np.testing.assert_array_equal(comp, G["mcc_out"].astype(np.int64))


def test_oracle_density_and_reflect_vs_reference(oracle, G):
    from oracle import render_ref as R
    # LearnedLaplaceDensity.density_func (models/rf/density.py:25-30) through get_alpha (intrinsic_avatar.py:390-394)
    beta = float(G["dens_beta"]) + float(G["dens_beta_min"])
    d = np.full_like(G["dens_sdf"], 0.02)
    alpha_ref = 1.0 - np.exp(-G["dens_out"].astype(np.float64) * 0.02)
    np.testing.assert_allclose(oracle.laplace_alpha(G["dens_sdf"], d, beta), alpha_ref, rtol=2e-6, atol=1e-7)
    # reflect (models/utils.py:115-116)
    np.testing.assert_allclose(R.reflect(G["reflect_x"], G["reflect_n"]), G["reflect_out"], rtol=1e-6, atol=1e-6)
