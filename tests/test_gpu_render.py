"""GPU: the whole render_step forward (BASELINE config 1 sizes: 128x128 frame, 64 samples/ray,
radiance only) against the CPU oracle restatement of the reference's forward_ on identical rays.

Floating-point tolerance: the field kernels use fma / MFMA and device exp/log, the oracle libm; SDF
values differ by ~1e-6 relative, so a CDF comparison inside the importance resampling can flip for a
vanishing fraction of rays; and the analytic normal is piecewise constant per hash cell, so a sample within
rounding distance of a cell face can get the neighbouring cell's normal (-> a different radiance input).
Stated bar (round 4): sample counts per ray identical (at most 2 rays may differ), every map held to (max, p99, mean) of its
per-pixel absolute difference at 3 x the MI355X observation, the maximum as a hard cap over all pixels (BARS_RENDER)."""
import numpy as np
import pytest
import torch

DEV = "cuda:0"

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame():
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S
    return S.build_frame("cuda:0", 128, 128, pose_seed=0, beta=0.01, num_samples_per_ray=64, grid_D=16, grid_H=64,
                         grid_W=64, smooth_iters=5, hash_amp=2e-3)


# observed on the MI355X (max / p99 / mean): comp_rgb 1.06e-2 / 5.2e-5 / 6.2e-6, opacity 4.1e-5 / 2.0e-6 / 1.1e-7, comp_normal 8.0e-2 /
# 2.5e-4 / 3.3e-5, depth 2.9e-5 / 2.4e-6 / 2.8e-7.  The maxima of comp_rgb / comp_normal are ONE pixel each: the analytic normal is
# piecewise constant per hash cell and a sample within an ulp of a cell face takes the neighbouring cell's normal (p99 is 200 x lower)
BARS_RENDER = {"comp_rgb": (3.2e-2, 1.6e-4, 1.9e-5), "opacity": (1.3e-4, 6.0e-6, 3.3e-7), "comp_normal": (0.25, 7.5e-4, 1.0e-4),
               "depth": (9.0e-5, 7.2e-6, 8.6e-7)}


def test_render_step_vs_oracle(frame, oracle):
    from oracle import render_ref as R
    rs, rays, export = frame
    out = rs.forward(rays)
    sc = R.Scene(**export)
    ref = R.render_step(sc, rays.cpu().numpy())
    st, sr = out["stats"], ref["stats"]
    assert sr["n_samples0"] > 5000
    assert st["n_edges0"] == sr["n_edges0"] and st["n_samples0"] == sr["n_samples0"]      # marching: bit-exact
    cnt = out["packed_info"][:, 1].cpu().numpy()
    cnt_ref = ref["packed_info"][:, 1]
    assert (cnt == cnt_ref).mean() >= 0.995, (cnt != cnt_ref).sum()
    # (max, p99, mean) of the per-pixel absolute difference: 3 x the MI355X observation (printed with -s), the max a hard cap over all
    # 16 384 pixels; sample counts per ray: observed identical, at most 2 rays may differ
    assert int((cnt != cnt_ref).sum()) <= 2, int((cnt != cnt_ref).sum())
    bad = []
    for k, bar in BARS_RENDER.items():
        a, b = out[k].cpu().numpy(), ref[k]
        err = np.abs(a - b).max(-1)
        got = (float(err.max()), float(np.quantile(err, 0.99)), float(err.mean()))
        print("render_step vs oracle", k, got)
        if not (got[0] <= bar[0] and got[1] <= bar[1] and got[2] <= bar[2]):
            bad.append((k, got, bar))
    assert not bad, bad
    hit = ref["opacity"][:, 0] > 0.5
    assert 0.02 < hit.mean() < 0.9
    # the one-pixel maxima of comp_rgb / comp_normal, DEMONSTRATED (tests/forward_golden.explain_gradient_outliers): with identical sample
    # sets, for every sample whose SDF gradient is more than 10 x p99 away from the oracle's it is the same sample (|shift| <= 1e-3) and
    # the HIP field AT THE ORACLE'S point returns the oracle's gradient (or the other side of a cell face a few ulp away does): position, not kernels;
    # the outlier pixels of comp_normal are the rays of those samples
    if int((cnt != cnt_ref).sum()) == 0:
        from tests import forward_golden as FG
        g_gpu, g_ref = out["sdf_grad"].cpu().numpy(), ref["sdf_grad"]
        both = out["valid"].cpu().numpy()
        e = np.where(both, np.abs(g_gpu - g_ref).max(-1), 0.0)
        thresh = 10.0 * float(np.quantile(e, 0.99))
        idx = np.nonzero(e > thresh)[0]
        r_smpl = rs.deformer.transform_rays_w2s(rays.float())
        ri = out["ray_indices"].long()
        sel_ = torch.from_numpy(idx).to(ri.device)
        dev = ri.device
        mid_g = (out["t_starts"] + out["t_ends"]) / 2.0
        mid_r = (torch.from_numpy(ref["t_starts"]).to(dev) + torch.from_numpy(ref["t_ends"]).to(dev)) / 2.0
        pts_g = (r_smpl[ri, :3] + r_smpl[ri, 3:6] * mid_g[:, None])[sel_]
        pts_r = (r_smpl[ri, :3] + r_smpl[ri, 3:6] * mid_r[:, None])[sel_]
        explained, why = FG.explain_gradient_outliers(rs, pts_g, pts_r, g_ref[idx], thresh)
        assert explained.all(), (idx[~explained].tolist(), {k: v[~explained].tolist() for k, v in why.items()}, thresh)
        flip_rays = set(ri.cpu().numpy()[idx].tolist())
        en = np.abs(out["comp_normal"].cpu().numpy() - ref["comp_normal"]).max(-1)
        bad_px = set(np.nonzero(en > 10.0 * float(np.quantile(en, 0.99)))[0].tolist())
        assert bad_px <= flip_rays, sorted(bad_px - flip_rays)
        print(f"cell-face check: {idx.size} gradient outliers of {e.size} samples, all explained")
    # compositing invariants
    op = out["opacity"][:, 0]
    assert float(op.min()) >= 0 and float(op.max()) <= 1 + 1e-5


def test_deform_vs_oracle(frame, oracle):
    """SNARFDeformer.deform (multi-candidate search + SDF + min-select + normal push-forward) on random points."""
    from oracle import render_ref as R
    rs, rays, export = frame
    sc = R.Scene(**export)
    g = torch.Generator().manual_seed(3)
    lo, hi = torch.tensor(export["aabb"][:3]), torch.tensor(export["aabb"][3:])
    pts = (torch.rand((6000, 3), generator=g) * (hi - lo) + lo)
    r = R.deform(sc, pts.numpy(), with_grad=True, with_feature=True)
    # search to the end + K9 (spec_eps = 0): the candidate bookkeeping is the oracle's, exactly
    old = rs.deformer.spec_eps
    try:
        rs.deformer.spec_eps = 0.0
        d0 = rs.deformer.deform(pts.cuda(), rs.geometry, with_grad=True, with_feature=True)
    finally:
        rs.deformer.spec_eps = old
    assert d0["n_candidates"] == r["n_candidates"]                      # Broyden + filter: bit-exact masks
    np.testing.assert_array_equal(d0["valid"].cpu().numpy(), r["valid"])
    # product default (K9-consistent early filter): the same candidates (a differing point is ~1e-7 of a batch: none of 6000)
    assert rs.deformer.spec_eps > 0
    d = rs.deformer.deform(pts.cuda(), rs.geometry, with_grad=True, with_feature=True)
    assert d["n_candidates"] == d0["n_candidates"]
    assert torch.equal(d["pts_cano"], d0["pts_cano"]) and torch.equal(d["sdf"], d0["sdf"]) and torch.equal(d["sel"], d0["sel"])
    np.testing.assert_array_equal(d["valid"].cpu().numpy(), r["valid"])
    v = r["valid"]
    assert 0.1 < v.mean() < 0.99
    np.testing.assert_allclose(d["sdf"].cpu().numpy(), r["sdf"], rtol=1e-4, atol=2e-5)
    same = np.abs(d["pts_cano"].cpu().numpy() - r["pts_cano"]).max(-1) < 1e-6     # same winner (ties aside)
    assert same.mean() > 0.999
    np.testing.assert_allclose(d["feature"].cpu().numpy()[same], r["feature"][same], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(d["sdf_grad"].cpu().numpy()[same], r["sdf_grad"][same], rtol=2e-3, atol=2e-3)


def test_forward_is_bit_reproducible():
    """no atomics, no uninitialised reads anywhere on the forward path: two runs of render_step on the same inputs give
    bit-identical sample sets and images (scans instead of atomics for every compaction, ordered look-back in the
    traversal, deterministic candidate order)."""
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S
    rs, rays, _ = S.build_frame(DEV, 96, 96, pose_seed=2, beta=0.02, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64,
                                smooth_iters=3, hash_amp=1e-2)
    junk = torch.full((64 << 20,), float("nan"), device=DEV)       # poison the allocator's free blocks
    del junk
    a = rs.forward(rays)
    sa = rs.sample(rays)
    junk = torch.full((64 << 20,), 1e30, device=DEV)
    del junk
    b = rs.forward(rays)
    sb = rs.sample(rays)
    for k in (3, 4, 5):
        assert torch.equal(sa[k], sb[k]), k
    for k in ("comp_rgb", "comp_normal", "opacity", "depth"):
        assert torch.equal(a[k], b[k]), k


def test_ray_batch_sharding_invariance_full_size():
    """BASELINE configs[1] size (540x540, 128 samples/ray).  Rays are independent given the replicated parameters and
    per-frame grids, which is what the multi-GPU ray-batch sharding relies on: rendering the frame in three uneven
    shards gives bit-identical pixels to rendering it at once (no result depends on which other rays share a launch,
    a tile, a wave or a scan), and the per-shard sample counts add up."""
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S
    rs, rays, _ = S.build_frame(DEV, 540, 540, pose_seed=0, beta=0.01, num_samples_per_ray=128)
    n = rays.shape[0]
    assert n == 291600
    full = rs.forward(rays)
    cuts = [0, 100_003, 100_003 + 4096, n]
    parts = [rs.forward(rays[a:b].contiguous()) for a, b in zip(cuts[:-1], cuts[1:])]
    for k in ("comp_rgb", "comp_normal", "opacity", "depth"):
        assert torch.equal(full[k], torch.cat([p[k] for p in parts], 0)), k
    assert sum(p["stats"]["n_samples"] for p in parts) == full["stats"]["n_samples"]
    assert sum(p["stats"]["n_edges0"] for p in parts) == full["stats"]["n_edges0"]
    hit = full["opacity"][:, 0] > 0.5
    assert 0.05 < float(hit.float().mean()) < 0.6 and full["stats"]["n_samples"] > 2_000_000



def test_config1_static_neutral_pose_vs_oracle(oracle):
    """BASELINE configs[0]: single 128x128 frame, static SMPL neutral pose (all bone transforms identity), 64 samples/ray,
    radiance only, against the CPU restatement of forward_ end to end."""
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S
    from oracle import render_ref as R
    rs, rays, export = S.build_frame(DEV, 128, 128, pose_seed=None, beta=0.01, num_samples_per_ray=64, grid_D=16, grid_H=64,
                                     grid_W=64, smooth_iters=5, hash_amp=2e-3)
    tfs = rs.deformer.tfs[0]
    assert torch.allclose(tfs, torch.eye(4, device=DEV).expand_as(tfs), atol=1e-6)      # neutral pose
    out = rs.forward(rays)
    ref = R.render_step(R.Scene(**export), rays.cpu().numpy())
    assert out["stats"]["n_edges0"] == ref["stats"]["n_edges0"] and out["stats"]["n_samples0"] == ref["stats"]["n_samples0"]
    cnt, cnt_ref = out["packed_info"][:, 1].cpu().numpy(), ref["packed_info"][:, 1]
    assert (cnt == cnt_ref).mean() >= 0.995
    assert int((cnt != cnt_ref).sum()) <= 2
    for k, bar in BARS_RENDER.items():
        err = np.abs(out[k].cpu().numpy() - ref[k]).max(-1)
        got = (float(err.max()), float(np.quantile(err, 0.99)), float(err.mean()))
        print("config1 vs oracle", k, got)
        assert got[0] <= bar[0] and got[1] <= bar[1] and got[2] <= bar[2], (k, got, bar)
    assert 0.02 < (ref["opacity"][:, 0] > 0.5).mean() < 0.9
