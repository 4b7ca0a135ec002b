"""IntrinsicAvatarModel.forward (models/intrinsic_avatar.py:1653-1666) over RenderStep: system.model_forward -- evaluation = chunk_batch over
forward_ with the chunks' dicts moved to the host and concatenated, + beta; training = forward_train_ + beta."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_model_forward_eval_is_the_chunked_forward_and_training_is_forward_train():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from intrinsicavatar_amd import build
    build.build()
    from intrinsicavatar_amd import synthetic as S, fields, pbr, system
    from tests.test_gpu_relight_oracle import hdri
    rs, rays, export = S.build_frame(DEV, 32, 32, pose_seed=0, beta=0.01, num_samples_per_ray=64, grid_D=16, grid_H=64, grid_W=64,
                                     smooth_iters=5, hash_amp=1e-2)
    mat = fields.VolumeMaterial(seed=2).to(DEV)
    env = pbr.EnvironmentLightTensor(torch.from_numpy(hdri()).to(DEV))
    env.update_pdf()
    n, spp = rays.shape[0], 64
    g = torch.Generator().manual_seed(3)
    light_u, shuffle_u = torch.rand((spp, 3), generator=g).to(DEV), torch.rand((n, spp), generator=g).to(DEV)
    bg = torch.tensor([0.2, 0.4, 0.6], device=DEV)
    kw = dict(background_color=bg, global_illumination=True)
    whole = rs.forward_(rays, mat, env, spp, light_u, shuffle_u, **kw)
    chunk = 300
    out = system.model_forward(rs, rays, mat, env, spp, light_u, shuffle_u, ray_chunk=chunk, **kw)
    assert sorted(out) == sorted(list(whole) + ["beta"])
    n_chunks = -(-n // chunk)
    for k, v in out.items():
        if k == "beta":
            assert torch.equal(v.detach().cpu(), rs.density.get_beta().detach().cpu())
            continue
        assert v.device.type == "cpu", k                                   # chunk_batch(..., move_to_cpu=True, ...)
        w = whole[k].cpu()
        if k.startswith("num_samples"):
            assert v.shape[0] == n_chunks and int(v.sum()) == int(w.sum()), (k, v, w)      # one entry per chunk, like the reference's cat
            continue
        assert v.shape == w.shape, (k, v.shape, w.shape)
        if v.dtype == torch.bool:
            assert int((v != w).sum()) <= 2, k
        else:
            # chunk invariance: the kernels a batch size selects differ in the last float bits, which can move a threshold decision of a
            # handful of Monte-Carlo samples (tests/test_gpu_relight_oracle.py: same statement for relight)
            d = (v - w).abs().reshape(n, -1).max(-1)[0]
            assert float((d > 1e-5).float().mean()) < 5e-3 and float(d.max()) < 5e-2, (k, float(d.max()), int((d > 1e-5).sum()))
    same = system.model_forward(rs, rays, mat, env, spp, light_u, shuffle_u, ray_chunk=n, move_to_cpu=False, **kw)
    for k, v in whole.items():
        assert torch.equal(same[k], v), k                                  # one chunk = forward_ itself, on the device
    # training form: forward_train_'s dict + beta, on the device, differentiable
    tr = system.model_forward(rs, rays[:256], mat, env, 16, torch.rand((256 * 16, 3), generator=g).to(DEV), None, training=True,
                              background_color=bg, global_illumination=True)
    assert "beta" in tr and tr["comp_rgb"].is_cuda and tr["comp_rgb"].requires_grad
