"""CPU: pin the oracle (oracle/ia_oracle*.c) against golden vectors produced by the
reference's own kernel bodies / modules (tests/golden/make_golden.py).

Integer / bool outputs: bit-exact.  Float outputs: bit-exact too for K1..K10 (same
IEEE operations in the same order, no FMA contraction on either side)."""
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_k1_k3_k4_resampling(oracle, golden_dir):
    g = _load(golden_dir, "golden_resampling.npz")
    for c in range(int(g["n_cases"])):
        k = f"c{c}_"
        pi, st, en = g[k + "packed_info"], g[k + "starts"], g[k + "ends"]
        w, al, sd, n = g[k + "weights"], g[k + "alphas"], g[k + "sdfs"], int(g[k + "n"])
        r = oracle.ray_resampling(pi, st, en, w, sd, n)
        for nm, v in zip(("rpi", "ts", "offsets", "indices", "fg_counts", "bg_counts", "surface_idx"), r):
            ref = g[k + "k1_" + nm]
            assert v.shape == ref.shape and v.dtype == ref.dtype, (c, nm, v.shape, ref.shape, v.dtype, ref.dtype)
            np.testing.assert_array_equal(v, ref, err_msg=f"K1 case {c} {nm}")
        r = oracle.ray_resampling_fine(pi, st, en, w, n)
        for nm, v in zip(("rpi", "starts", "ends", "is_fg"), r):
            np.testing.assert_array_equal(v, g[k + "k3_" + nm], err_msg=f"K3 case {c} {nm}")
        r = oracle.ray_resampling_sdf_fine(pi, st, en, al, sd, n)
        for nm, v in zip(("rpi", "starts", "ends", "is_fg"), r):
            np.testing.assert_array_equal(v, g[k + "k4_" + nm], err_msg=f"K4 case {c} {nm}")


def test_k2_resampling_merge(oracle, golden_dir):
    g = _load(golden_dir, "golden_resampling.npz")
    for c in range(int(g["n_edge_cases"])):
        k = f"e{c}_"
        r = oracle.ray_resampling_merge(g[k + "packed_info"], g[k + "vals"], g[k + "is_left"], g[k + "is_right"],
                                        g[k + "weights"], int(g[k + "n"]))
        for nm, v in zip(("rpi", "vals", "dists", "is_left", "is_right", "is_resample", "is_fg"), r):
            np.testing.assert_array_equal(v, g[k + "k2_" + nm], err_msg=f"K2 case {c} {nm}")


def test_k5_k7_pack(oracle, golden_dir):
    g = _load(golden_dir, "golden_pack.npz")
    pi, data = g["packed_info"], g["data"]
    np.testing.assert_array_equal(oracle.unpack_info(pi, data.shape[0]), g["unpack_info"])
    np.testing.assert_array_equal(oracle.unpack_info_to_mask(pi, 32), g["unpack_mask"])
    np.testing.assert_array_equal(oracle.unpack_data(pi, data, 32), g["unpack_data"])
    # pack_info is the inverse of unpack_info (lib/nerfacc/pack.py:46-77)
    np.testing.assert_array_equal(oracle.pack_info(g["unpack_info"], pi.shape[0]), pi)


def test_k8_k10_fast_snarf(oracle, golden_dir):
    g = _load(golden_dir, "golden_snarf.npz")
    vw = g["voxel_w"].astype(np.float32)
    vd, vJ = oracle.precompute(vw, g["tfs"], g["offset"], g["scale"])
    np.testing.assert_array_equal(vd, g["voxel_d"])
    np.testing.assert_array_equal(vJ, g["voxel_J"])
    x, Ji, valid = oracle.fuse_broyden(g["xd"], g["voxel_J"], g["tfs"], g["bones"], g["offset"], g["scale"])
    np.testing.assert_array_equal(valid, g["valid"])
    np.testing.assert_array_equal(x, g["x"])
    np.testing.assert_array_equal(Ji, g["J_inv"])
    assert 0 < valid.sum() < valid.size
    np.testing.assert_array_equal(oracle.filter(x, valid), g["filtered"])
    assert g["filtered"].sum() < valid.sum()


def _weight_norm(g, v):
    return g * v / np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True).astype(np.float32)


def test_mlps_against_reference_modules(oracle, golden_dir):
    g = _load(golden_dir, "golden_mlp.npz")
    # SDF net (VanillaMLP, weight-norm, Softplus beta=100): models/network_utils.py:201-244
    W1 = _weight_norm(g["sdf_sd_layers.0.weight_g"], g["sdf_sd_layers.0.weight_v"])
    W2 = _weight_norm(g["sdf_sd_layers.2.weight_g"], g["sdf_sd_layers.2.weight_v"])
    y = oracle.mlp_fwd(g["sdf_x"], [W1, W2], [g["sdf_sd_layers.0.bias"], g["sdf_sd_layers.2.bias"]], "softplus100")
    np.testing.assert_allclose(y, g["sdf_y"], rtol=2e-5, atol=2e-6)
    # radiance net (VanillaMLP ReLU)
    y = oracle.mlp_fwd(g["rad_x"], [g[f"rad_sd_layers.{i}.weight"] for i in (0, 2, 4)],
                       [g[f"rad_sd_layers.{i}.bias"] for i in (0, 2, 4)], "relu")
    np.testing.assert_allclose(y, g["rad_y"], rtol=2e-5, atol=2e-6)
    # material net (LipshitzMLP): network_utils.py:396-428
    Ws, bs = [], []
    for i in range(3):
        w = g[f"mat_sd_weights_per_layer.{i}"]
        c = g[f"mat_sd_lipshitz_bound_per_layer.{i}"]
        sp = np.log1p(np.exp(c.astype(np.float64))).astype(np.float32) if c < 20 else c
        scale = np.minimum(sp / np.abs(w).sum(1), 1.0).astype(np.float32)
        Ws.append(w * scale[:, None])
        bs.append(g[f"mat_sd_biases_per_layer.{i}"])
    assert any((np.minimum(1, 1) and (np.abs(W).sum(1) < np.abs(g[f"mat_sd_weights_per_layer.{i}"]).sum(1) - 1e-6).any())
               for i, W in enumerate(Ws)), "Lipschitz clamp should be active in the fixture"
    y = oracle.mlp_fwd(g["mat_x"], Ws, bs, "relu")
    np.testing.assert_allclose(y, g["mat_y"], rtol=2e-5, atol=2e-6)
