"""GPU (MI355X): the hot path on frames of the reference's OWN pose files (tests/golden/reference_poses.npz: four training frames of
load/peoplesnapshot/male-3-casual/poses/anim_nerf_train.npz, four out-of-distribution frames of load/animation/aist/poses.npz,
translation re-based as datasets/animation.py:129-130) driven through plain forward kinematics -- BASELINE configs 2-5 name these
files.  Bars: (max, p99, mean) per key over EVERY pixel, 3 x the MI355X observation under the hard caps written here
(tests/parity_bars.py); discrete outputs exact up to a stated number of flips."""
import numpy as np
import pytest
import torch

from tests import parity_bars as PB

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def _frame(pose, hw):
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S
    return S.build_frame(DEV, hw, hw, pose=pose, beta=0.01, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64, smooth_iters=5,
                         hash_amp=2e-3)


def test_pose_fixture_and_fk_against_smpl_py():
    """the committed frames load, and the package's SMPL forward kinematics (smpl.py, pinned against the reference's smplx/lbs.py)
    reproduces the rig transforms the scene builder derives from them."""
    from intrinsicavatar_amd import synthetic as S, smpl
    for name, frames in (("male-3-casual", (0, 40, 80, 113)), ("aist", (0, 100, 200, 319))):
        for f in frames:
            pose, transl = S.reference_pose(name, f)
            assert pose.shape == (72,) and np.isfinite(pose).all() and 3.0 < transl[2] < 7.0
            rig = S.make_rig(pose, transl)
            J = T(S.JOINTS)[None]
            eye = torch.eye(24, device=DEV)
            body = smpl.SMPLKinematics(J, torch.zeros((24, 3, 1), device=DEV), torch.zeros((207, 72), device=DEV), eye, S.PARENTS.tolist(), eye)
            p = T(pose.astype(np.float32))[None]
            out = body.forward(torch.zeros((1, 1), device=DEV), p[:, 3:], p[:, :3], T(transl.astype(np.float32))[None])
            tfs, w2s = smpl.deformer_transforms(out["A"], torch.eye(4, device=DEV).expand(1, 24, 4, 4))
            np.testing.assert_allclose(N(tfs), rig["tfs"], atol=5e-5)
            np.testing.assert_allclose(N(w2s[0]), rig["w2s"], atol=5e-5)


@pytest.mark.parametrize("pose", ["male-3-casual:40", "aist:200"])
def test_render_step_vs_oracle_on_reference_poses(oracle, pose):
    """BASELINE config 2 shape (radiance + SDF geometry) on a peoplesnapshot training frame and an animation frame."""
    from oracle import render_ref as R
    rs, rays, export = _frame(pose, 96)
    # the runtime canary of the search's early filter (SNARFDeformer.spec_canary) rides along: every 16th point of every search
    # batch of this frame is searched again to the end + K9 (filter.cu:10-54) and compared with the row the product search left
    rs.deformer.spec_canary = 16
    rs.deformer.canary_totals(reset=True)
    out = rs.forward(rays)
    n_chk, n_bad, _ = rs.deformer.canary_totals(reset=True)
    rs.deformer.spec_canary = 0
    assert n_chk > 5000 and n_bad == 0, (n_chk, n_bad)
    ref = R.render_step(R.Scene(**export), N(rays))
    st, sr = out["stats"], ref["stats"]
    assert sr["n_samples0"] > 3000
    assert st["n_edges0"] == sr["n_edges0"] and st["n_samples0"] == sr["n_samples0"]      # marching: bit-exact
    cnt, cnt_ref = N(out["packed_info"][:, 1]), ref["packed_info"][:, 1]
    PB.count(f"refpose/{pose}/rays_with_another_sample_count", int((cnt != cnt_ref).sum()), 2)
    # hard caps: one sample on the other side of a hash-cell face moves ONE pixel's normal by ~0.1 and its colour by ~1e-2
    for k, cap in (("comp_rgb", (3e-2, 1.5e-3, 1e-4)), ("opacity", (4e-4, 3e-5, 2e-6)), ("comp_normal", (0.25, 5e-3, 1e-3)),
                   ("depth", (3e-4, 3e-5, 3e-6))):
        PB.held(f"refpose/{pose}/{k}", N(out[k]), ref[k], cap)
    # the one-pixel maxima of comp_normal, DEMONSTRATED sample by sample (tests/forward_golden.explain_gradient_outliers) when the sample sets coincide
    if int((cnt != cnt_ref).sum()) == 0:
        from tests import forward_golden as FG
        print(f"refpose/{pose}: {FG.assert_normal_outliers_explained(rs, rays, out, ref)} gradient outliers, all explained")


def test_relight_vs_oracle_on_an_animation_pose(oracle):
    """BASELINE config 5 (animation OOD pose, render_mode=light, samples_per_pixel=1024, global illumination on) on an oracle-sized
    subsample of the frame's rays (24 x 24; the oracle marches ~0.6 M secondary rays for them)."""
    from intrinsicavatar_amd import synthetic as S, fields, pbr
    from oracle import render_ref as R
    from tests.test_gpu_relight_oracle import hdri
    rs, rays, export = _frame("aist:100", 24)
    mat = fields.VolumeMaterial(seed=2).to(DEV)
    env = pbr.EnvironmentLightTensor(T(hdri()))
    env.update_pdf()
    sc = R.Scene(**export, **S.export_phys(mat, env.base))
    n, spp = rays.shape[0], 1024
    rng = np.random.default_rng(7)
    light_u, shuffle_u = rng.random((spp, 3), dtype=np.float32), rng.random((n, spp), dtype=np.float32)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    ref = R.relight_step(sc, N(rays), spp=spp, light_u=light_u, shuffle_u=shuffle_u, global_illumination=True, background_color=bg)
    rs.deformer.spec_canary = 1024                       # the canary over the ~0.6 M secondary rays' march points as well: must count nothing
    rs.deformer.canary_totals(reset=True)
    d = rs.forward_(rays, mat, env, spp, T(light_u), T(shuffle_u), background_color=T(bg), global_illumination=True)
    n_chk, n_bad, _ = rs.deformer.canary_totals(reset=True)
    rs.deformer.spec_canary = 0
    assert n_chk > 1000 and n_bad == 0, (n_chk, n_bad)
    want = R.forward_output_dict(ref, bg, "light")
    assert sorted(d) == sorted(want)
    assert ref["stats"]["n_fg"] > 50_000
    PB.count("refpose/aist:100/num_samples_diff", abs(int(d["num_samples"][0]) - int(want["num_samples"][0])), 2)
    for k, cap in (("comp_rgb", (3e-2, 1.5e-3, 1e-4)), ("comp_albedo", (3e-4, 2e-5, 1e-6)), ("opacity", (4e-4, 3e-5, 2e-6)),
                   ("comp_rgb_full", (3e-2, 1.5e-3, 1e-4)), ("comp_normal", (0.25, 5e-3, 1e-3))):
        PB.held(f"refpose/aist:100/relight/{k}", N(d[k]), want[k], cap)
    # Monte-Carlo images (spp 1024): a visibility sample on the other side of a threshold moves a pixel by Lo / spp
    for k, cap in (("comp_rgb_phys", (0.1, 1e-2, 5e-4)), ("comp_demod_phys", (0.3, 3e-2, 1.5e-3)), ("comp_rgb_phys_full", (0.1, 1e-2, 5e-4))):
        PB.held(f"refpose/aist:100/relight/{k}", N(d[k]), want[k], cap)
