"""SMPL forward kinematics (intrinsicavatar_amd/smpl.py) against golden vectors produced by the reference's own
models/deformers/smplx/lbs.py (tests/golden/make_golden_smpl.py) -- runs on CPU, no GPU involved."""
import os

import numpy as np
import torch

from intrinsicavatar_amd import smpl

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_smpl.npz"))


def _model(dt):
    t = lambda k: torch.from_numpy(G[k]).to(dt)      # noqa: E731
    return smpl.SMPLKinematics(t("v_template"), t("shapedirs"), t("posedirs"), t("J_regressor"), G["parents"].tolist(),
                               t("lbs_weights"))


def test_rodrigues_matches_reference():
    for dt, tag, tol in ((torch.float64, "f64", 1e-14), (torch.float32, "f32", 1e-6)):
        R = smpl.rodrigues(torch.from_numpy(G["pose"]).to(dt).reshape(-1, 3))
        np.testing.assert_allclose(R.numpy(), G[f"rodrigues_{tag}"], rtol=0, atol=tol)
    R = smpl.rodrigues(torch.zeros(1, 3))
    assert torch.allclose(R[0], torch.eye(3), atol=1e-7)


def test_lbs_matches_reference():
    for dt, tag, tol in ((torch.float64, "f64", 1e-12), (torch.float32, "f32", 5e-6)):
        m = _model(dt)
        pose = torch.from_numpy(G["pose"]).to(dt)
        out = m.forward(torch.from_numpy(G["betas"]).to(dt), pose[:, 3:], pose[:, :3])
        np.testing.assert_allclose(out["A"].numpy(), G[f"A_{tag}"], rtol=0, atol=tol)
        np.testing.assert_allclose(out["joints"].numpy(), G[f"joints_{tag}"], rtol=0, atol=tol)
        np.testing.assert_allclose(out["vertices"].numpy(), G[f"verts_{tag}"], rtol=0, atol=tol)
    # translation moves joints, vertices and the bone transforms' translation column (body_models.py:350-358)
    tr = torch.from_numpy(G["transl"])
    m = _model(torch.float64)
    pose = torch.from_numpy(G["pose"])
    o2 = m.forward(torch.from_numpy(G["betas"]), pose[:, 3:], pose[:, :3], tr)
    np.testing.assert_allclose(o2["joints"].numpy(), G["joints_f64"] + G["transl"][:, None], atol=1e-12)
    np.testing.assert_allclose(o2["A"][:, :, :3, 3].numpy(), G["A_f64"][:, :, :3, 3] + G["transl"][:, None], atol=1e-12)


def test_rest_pose_and_pose_gradient():
    m = _model(torch.float64)
    betas = torch.from_numpy(G["betas"])[:1]
    rest = m.forward(betas, torch.zeros(1, 69, dtype=torch.float64), torch.zeros(1, 3, dtype=torch.float64))
    eye = torch.eye(4, dtype=torch.float64).expand(24, 4, 4)
    assert torch.allclose(rest["A"][0], eye, atol=1e-7)            # zero pose: every bone transform is the identity
    A_rest_inv = torch.linalg.inv(rest["A"])
    pose = torch.from_numpy(G["pose"])[:1].clone().requires_grad_(True)
    out = m.forward(betas, pose[:, 3:], pose[:, :3])
    tfs, w2s = smpl.deformer_transforms(out["A"], A_rest_inv, dtype=torch.float64)
    assert torch.allclose(tfs[0, 0], torch.eye(4, dtype=tfs.dtype), atol=1e-6)   # root bone == SMPL frame
    # d (sum tfs * c) / d pose against central differences
    c = torch.randn(tfs.shape, generator=torch.Generator().manual_seed(0), dtype=tfs.dtype)
    (tfs * c).sum().backward()
    gnum = torch.zeros_like(pose)
    with torch.no_grad():
        for k in range(0, 72, 7):
            e = torch.zeros_like(pose); e[0, k] = 1e-6
            f = lambda p: (smpl.deformer_transforms(m.forward(betas, p[:, 3:], p[:, :3])["A"], A_rest_inv, dtype=torch.float64)[0] * c).sum()   # noqa: E731
            gnum[0, k] = (f(pose + e) - f(pose - e)) / 2e-6
            assert abs(float(gnum[0, k] - pose.grad[0, k])) < 1e-4 * max(1.0, abs(float(gnum[0, k]))), k
