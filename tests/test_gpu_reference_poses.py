"""GPU (MI355X): the hot path on frames of the reference's OWN pose files (tests/golden/reference_poses.npz: four training frames of
load/peoplesnapshot/male-3-casual/poses/anim_nerf_train.npz, four out-of-distribution frames of load/animation/aist/poses.npz,
translation re-based as datasets/animation.py:129-130) driven through plain forward kinematics -- BASELINE configs 2-5 name these
files.  Same bars as the synthetic-pose tests (tests/test_gpu_render.py, tests/test_gpu_relight_oracle.py)."""
import numpy as np
import pytest
import torch

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
    out = rs.forward(rays)
    ref = R.render_step(R.Scene(**export), N(rays))
    st, sr = out["stats"], ref["stats"]
    assert sr["n_samples0"] > 3000
    assert st["n_edges0"] == sr["n_edges0"] and st["n_samples0"] == sr["n_samples0"]      # marching: bit-exact
    cnt, cnt_ref = N(out["packed_info"][:, 1]), ref["packed_info"][:, 1]
    assert (cnt == cnt_ref).mean() >= 0.995
    for k, tol in (("comp_rgb", 2e-3), ("opacity", 2e-3), ("comp_normal", 4e-3), ("depth", 5e-3)):
        err = np.abs(N(out[k]) - ref[k]).max(-1)
        assert (err < tol).mean() >= 0.985 and err.max() < 0.15 and err.mean() < 2e-4, (k, float(err.max()), float(err.mean()))


def test_relight_vs_oracle_on_an_animation_pose(oracle):
    """BASELINE config 5 shape (animation pose, render_mode=light, global illumination on) at an oracle-sized frame."""
    from intrinsicavatar_amd import synthetic as S, fields, pbr
    from oracle import render_ref as R
    from tests.test_gpu_relight_oracle import hdri
    rs, rays, export = _frame("aist:100", 32)
    mat = fields.VolumeMaterial(seed=2).to(DEV)
    env = pbr.EnvironmentLightTensor(T(hdri()))
    env.update_pdf()
    sc = R.Scene(**export, **S.export_phys(mat, env.base))
    n, spp = rays.shape[0], 64
    rng = np.random.default_rng(7)
    light_u, shuffle_u = rng.random((spp, 3), dtype=np.float32), rng.random((n, spp), dtype=np.float32)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    ref = R.relight_step(sc, N(rays), spp=spp, light_u=light_u, shuffle_u=shuffle_u, global_illumination=True, background_color=bg)
    d = rs.forward_(rays, mat, env, spp, T(light_u), T(shuffle_u), background_color=T(bg), global_illumination=True)
    want = R.forward_output_dict(ref, bg, "light")
    assert sorted(d) == sorted(want)
    assert ref["stats"]["n_fg"] > 1000
    assert abs(int(d["num_samples"][0]) - int(want["num_samples"][0])) <= 0.005 * int(want["num_samples"][0])
    for k, tol in (("comp_rgb", 2e-3), ("comp_albedo", 2e-3), ("opacity", 2e-3), ("comp_rgb_full", 4e-3)):
        err = np.abs(N(d[k]) - want[k]).max(-1)
        assert (err < tol).mean() >= 0.985, (k, float(err.max()))
    for k in ("comp_rgb_phys", "comp_demod_phys", "comp_rgb_phys_full"):
        a, b = N(d[k]), want[k]
        tol = 2e-2 * np.abs(b).max(-1) + 2e-2
        assert (np.abs(a - b).max(-1) <= tol).mean() >= 0.97, k
