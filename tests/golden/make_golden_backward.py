#!/usr/bin/env python3
"""Golden GRADIENTS of the whole hot path: the reference's own training step run on CPU in the build container --
`IntrinsicAvatarSystem.training_step` (/root/reference/systems/intrinsic_avatar.py:160-301, imported by file path, called
unbound on a stand-in `self`) on the reference's own `IntrinsicAvatarModel.forward` in train() mode (the scene, rays and
random tensors of tests/golden/golden_forward.npz's `light_16_gi_train` run), then `loss.backward()` through the reference's
autograd graph:

    models/rf/geometry.py:165-172      sdf normal by autograd.grad(create_graph=True) -> eikonal / normal terms double-backward
    models/volrend.py:810-1020         rendering_with_normals_mats_sdf
    models/intrinsic_avatar.py         rgb_normal_mats_alpha_fn incl. the material jitter pass, volume scattering, pbr_light_forward
    models/pbr/utils.py                sample_volume_interaction gathers, lib/nerfacc/pack.py unpack_data (autograd Function)
    models/pbr/material.py:53-87       regularizations (smoothness / orientation means, Gaussian-histogram entropy)
    models/network_utils.py            weight-normed VanillaMLP (weight_g / weight_v), LipshitzMLP (+ lipshitz_bound regulariser)

  python tests/golden/make_golden_backward.py          (needs /root/reference; CPU only; ~5 minutes)

What the reference tree has no CPU implementation of is supplied test-side exactly as in make_golden_forward.py; for the
backward those stand-ins must be differentiable, so during this run
    nerfacc.render_weight_from_alpha / accumulate_along_rays   VALUES from the CPU oracle (as before), GRADIENT from a torch
                                                               restatement (T_i = prod_{j<i}(1 - a_j); index_add)
    lib.torch_pbr  MultiLobe.eval / EnvironmentLightTensor.eval VALUES from oracle/pbr_ref.py, GRADIENT from tests/torch_ref.py
                                                               (brdf_eval_t / env_eval_t, fp64)
    tinycudann     tests/torch_ref.hashgrid (plain torch, differentiable to any order)
via `value + (twin - twin.detach())`, which adds an exact zero: the forward values of this run are bit-identical to the
`light_16_gi_train` arrays of golden_forward.npz (asserted below).  nerfacc / tinycudann / torch_pbr are third-party and
absent, so their own backward is still "vs our restatement" -- what this fixture pins is everything of the REFERENCE's that
sits between them: its graph, its losses, its parameterisations.

Three loss compositions, one backward each (same scene, same random tensors), on the `light` estimator at spp 16 -- and a fourth
run, `uniform_default`, on the estimator the reference SHIPS for training (render_mode uniform_light, samples_per_pixel 512:
configs/config.yaml:46-48, models/intrinsic_avatar.py:654-753,1392-1413; the `uniform_light_512_gi_train` run of golden_forward.npz)
with the default composition:
    default   configs/config.yaml:87-109 as shipped (rgb L1 1, phys L1 0.2, mask BCE 0.1, eikonal 0.1, Lipschitz 1e-5 at
              step 25 000, the three smoothness terms 0.01)
    allterms  every term of training_step that the default sets to zero switched on as well (MSEs, demodulated, mask MSE,
              opaque, sparsity, normal orientation, albedo entropy), smoothness weights raised so they are not lost in fp32
    lipshitz  default + phys MSE, with the material MLP's Lipschitz bounds scaled by 0.3 so that the weight normalisation is
              active (d loss / d bound != 0); its forward therefore differs from golden_forward.npz in the material maps, which
              are stored

Only DATA is written (tests/golden/golden_backward.npz): targets, loss values, and d loss / d parameter for every parameter
group.  The two 50 MB hash-table gradients are stored as: per-level sums / L1 / L2 norms over ALL entries, a projection on a
closed-form pseudo-random vector, and the exact values of the touched entries whose index passes a 1-in-SUBSAMPLE hash.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

import make_golden_forward as GF          # noqa: E402
from intrinsicavatar_amd import synthetic as S      # noqa: E402

N = GF.N
Cfg = GF.Cfg
from tests.forward_golden import SUBSAMPLE      # noqa: E402
TAG = "light_16_gi_train"
UNIFORM_TAG = "uniform_light_512_gi_train"      # the estimator the reference ships for training (configs/config.yaml:46-48)

LOSS_DEFAULT = dict(lambda_rgb_l1=1.0, lambda_rgb_phys_l1=0.2, lambda_mask_bce=0.1, lambda_eikonal=0.1,
                    lambda_lipshitz_bound=[12500, 1.0e-5, 1.0e-5, 12501], lambda_curvature=[1.5, 0.0, 12500],
                    lambda_albedo_smoothness=0.01, lambda_roughness_smoothness=0.01, lambda_metallic_smoothness=0.01,
                    sparsity_scale=1.0, lambda_rgb_mse=0.0, lambda_rgb_phys_mse=0.0, lambda_rgb_demodulated=0.0, lambda_mask_mse=0.0,
                    lambda_sparsity=0.0, lambda_distortion=0.0, lambda_opaque=0.0, lambda_albedo=0.0, lambda_normal_orientation=0.0,
                    lambda_albedo_entropy=0.0, lambda_energy_conservation=0.0)
LOSS_ALLTERMS = dict(LOSS_DEFAULT, lambda_rgb_mse=0.5, lambda_rgb_phys_mse=0.5, lambda_rgb_demodulated=0.1, lambda_mask_mse=0.1,
                     lambda_sparsity=0.01, lambda_opaque=0.01, lambda_normal_orientation=0.05, lambda_albedo_entropy=0.001,
                     lambda_albedo_smoothness=1.0e6, lambda_roughness_smoothness=1.0e6, lambda_metallic_smoothness=1.0e6,
                     lambda_lipshitz_bound=[12500, 1.0e-3, 1.0e-3, 12501])
VARIANTS = dict(default=LOSS_DEFAULT, allterms=LOSS_ALLTERMS, lipshitz=dict(LOSS_DEFAULT, lambda_rgb_phys_mse=0.5))
# lipshitz only: the Lipschitz bounds of the material MLP are initialised at 2 x the largest row sum (network_utils.py:380-385), where
# the normalisation (:396-403) is the identity and d / d bound = 0; scaled down so that the clamp is active on the large rows
LIPSHITZ_SCALE = dict(default=1.0, allterms=1.0, lipshitz=0.3, uniform_default=1.0)
# uniform_default: the SHIPPED training configuration -- render_mode uniform_light, samples_per_pixel 512 (configs/config.yaml:46-48,
# BASELINE configs[3]) with the default loss composition, on the `uniform_light_512_gi_train` run of golden_forward.npz.  Variants are
# only ever appended (the arrays of the earlier ones must regenerate bit-identically).
VARIANTS["uniform_default"] = LOSS_DEFAULT
RUN_OF = dict(default=(TAG, "light", 16), allterms=(TAG, "light", 16), lipshitz=(TAG, "light", 16), uniform_default=(UNIFORM_TAG, "uniform_light", 512))


from tests.forward_golden import table_gradient_summary      # noqa: E402  (shared with the GPU test)


# ----------------------------------------------------------------------------- differentiable stand-ins (value-preserving)
def _with_grad(value, twin):
    """`value` (from the numpy oracle) with the gradient of `twin` (torch restatement): adds an exact zero."""
    return value + (twin - twin.detach()).to(value.dtype)


def patch_everywhere(old, new):
    for m in list(sys.modules.values()):
        d = getattr(m, "__dict__", None)
        if not d:
            continue
        for k, v in list(d.items()):
            if v is old:
                d[k] = new


def make_differentiable_shims():
    from tests import torch_ref as TR
    nf = sys.modules["nerfacc"]
    rw0, acc0 = nf.render_weight_from_alpha, nf.accumulate_along_rays

    def render_weight_from_alpha(alphas, packed_info=None, ray_indices=None, n_rays=None, prefix_trans=None):
        w, tr = rw0(alphas, packed_info, ray_indices, n_rays)
        if not (torch.is_grad_enabled() and alphas.requires_grad):
            return w, tr
        if packed_info is None:
            packed_info = sys.modules["lib.nerfacc"].pack_info(ray_indices, n_rays)
        a = alphas.double()
        T = torch.ones_like(a)
        pieces = []
        for s, c in packed_info.tolist():
            if c > 0:
                om = 1.0 - a[s:s + c]
                pieces.append((s, torch.cat([om.new_ones(1), torch.cumprod(om, 0)[:-1]])))
        T = torch.cat([p for _, p in sorted(pieces, key=lambda t: t[0])]) if pieces else a
        assert T.shape == a.shape, "packed_info must cover the samples contiguously"
        return _with_grad(w, a * T), _with_grad(tr, T)

    def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
        out = acc0(weights, values, ray_indices, n_rays)
        need = torch.is_grad_enabled() and (weights.requires_grad or (values is not None and values.requires_grad))
        if not need:
            return out
        src = weights.double()[:, None] * (values.double() if values is not None else 1.0)
        twin = torch.zeros((int(n_rays), src.shape[1]), dtype=torch.float64).index_add_(0, ray_indices.long(), src)
        return _with_grad(out, twin)

    for old, new in ((rw0, render_weight_from_alpha), (acc0, accumulate_along_rays)):
        patch_everywhere(old, new)

    tp = sys.modules["lib.torch_pbr"]
    ev0, em0 = tp.MultiLobe.eval, tp.EnvironmentLightTensor.eval

    def brdf_eval(self, wi, n, wo, alpha_x, alpha_y, albedo, metallic, attenuation):
        d, s = ev0(self, wi, n, wo, alpha_x, alpha_y, albedo, metallic, attenuation)
        if not torch.is_grad_enabled():
            return d, s
        # rows where the specular lobe is defined: the twin is only evaluated there (no 0 * inf in its backward)
        dn, dwi, dwo = n.double(), wi.double(), wo.double()
        ok = ((dn * dwo).sum(-1) > 0) & ((dn * dwi).sum(-1) > 0) & ((dwi + dwo).norm(dim=-1) >= 1e-12)
        lit = (dn * dwo).sum(-1) > 0
        d_t = torch.where(lit, (dn * dwo).sum(-1) / np.pi, torch.zeros_like(lit, dtype=torch.float64))[:, None]
        s_t = torch.zeros((n.shape[0], 3), dtype=torch.float64)
        if bool(ok.any()):
            _, sp = TR.brdf_eval_t(dn[ok], dwi[ok], dwo[ok], alpha_x.double()[ok], albedo.double()[ok], metallic.double()[ok][:, 0])
            s_t = s_t.index_put((torch.nonzero(ok)[:, 0],), sp)
        return _with_grad(d, d_t), _with_grad(s, s_t)

    def env_eval(self, d):
        v = em0(self, d)
        if not (torch.is_grad_enabled() and self.base.requires_grad):
            return v
        return _with_grad(v, TR.env_eval_t(self.base.double(), d.detach().double()))

    tp.MultiLobe.eval, tp.EnvironmentLightTensor.eval = brdf_eval, env_eval
    tp.luma = lambda x: ((x[..., 0:1] + x[..., 1:2] + x[..., 2:3]) / 3.0).expand_as(x)
    tp.max_value = lambda x: torch.max(x, dim=-1, keepdim=True)[0].expand_as(x)


def import_reference_system():
    """systems/intrinsic_avatar.py by file path: what it imports beside the model is logging / metrics / Lightning."""
    MH = sys.modules["make_golden_host"]
    for name in ("wandb", "lpips", "torch_efficient_distloss", "skimage", "skimage.metrics", "utils.mixins", "systems.criterions"):
        sys.modules.setdefault(name, types.ModuleType(name))

    def _not_on_path(*a, **k):
        raise NotImplementedError("lambda_distortion = 0: never called")
    sys.modules["torch_efficient_distloss"].flatten_eff_distloss = _not_on_path
    sys.modules["skimage.metrics"].structural_similarity = _not_on_path
    pl = sys.modules["pytorch_lightning"]
    pl.LightningModule = type("LightningModule", (nn.Module,), {})
    sys.modules["utils.mixins"].SaverMixin = type("SaverMixin", (), {})
    su = sys.modules["systems.utils"]
    for k in ("parse_optimizer", "parse_scheduler"):
        setattr(su, k, _not_on_path)
    sy = sys.modules["systems"]
    sy.register = lambda name: (lambda cls: cls)
    import cv2 as _cv2                                           # noqa: F401   (the stub module make_golden_host registered)
    crit = MH.load("systems.criterions", f"{REF}/systems/criterions.py")           # binary_cross_entropy (:229-233) and the metrics
    MH.load("systems.base", f"{REF}/systems/base.py")
    sysm = MH.load("systems.intrinsic_avatar", f"{REF}/systems/intrinsic_avatar.py")
    assert sysm.binary_cross_entropy is crit.binary_cross_entropy
    return sysm


class _SystemStandIn:
    """the attributes IntrinsicAvatarSystem.training_step / BaseSystem.C read from `self`."""

    def __init__(self, sysm, model, loss_cfg):
        self.model = model
        self.config = Cfg(system=Cfg(pbr_loss_only=False, loss=Cfg(loss_cfg)), model=Cfg(learn_material=True, learned_background=False))
        self.global_step, self.current_epoch = 25000, 250
        self.dataset = types.SimpleNamespace(has_mask=True)
        self.train_num_rays = 0
        self.logged = {}
        self.out = None
        self._C = sysm.BaseSystem.C

    def C(self, value):
        return self._C(self, value)

    def log(self, name, value, **kw):
        self.logged[name] = float(value) if not isinstance(value, str) else value

    def __call__(self, batch):
        self.out = self.model(batch["rays"])                       # IntrinsicAvatarSystem.forward (:43-44)
        return self.out


def targets(G):
    """training targets of the frame (what preprocess_data leaves in the batch, :84-158): a smooth image and a soft mask."""
    n = G["rays"].shape[0]
    hw = int(round(np.sqrt(n)))
    v, u = np.meshgrid((np.arange(hw) + 0.5) / hw, (np.arange(hw) + 0.5) / hw, indexing="ij")
    rgb = np.stack([0.5 + 0.4 * np.sin(5 * u + 1), 0.5 + 0.4 * np.cos(4 * v), 0.5 + 0.3 * np.sin(3 * (u + v))], -1).reshape(n, 3)
    r = np.sqrt((u - 0.5) ** 2 + ((v - 0.5) / 1.6) ** 2).reshape(n)
    alpha = np.clip((0.28 - r) / 0.04, 0.0, 1.0)
    return rgb.astype(np.float32), alpha.astype(np.float32)


def build_model(mods, IA, rd, bg, hdri, mode="light", spp=16):
    model = IA.IntrinsicAvatarModel(GF.model_config(mode, spp, True))
    GF.init_params(model)
    model.eval()
    model.update_step(250, 25000)
    model.train(True)
    model.background_color = bg
    model.geometry.prepare_bbox(rd.bbox)
    model.radiance.prepare_bbox(rd.bbox)
    model.jitter_materials = True
    model.with_curvature_loss = False
    model.cond = None
    model.t_idx = 0.0
    model.occupancy_grid._update(step=0, t_idx=0, occ_eval_fn=lambda x: GF._occ_eval(model, x), occ_thre=0.001, ema_decay=0.8)
    model.emitter.base = nn.Parameter(torch.from_numpy(hdri))
    model.emitter.pdf_scale = (model.emitter.base.shape[0] * model.emitter.base.shape[1]) / (2 * np.pi * np.pi)
    model.emitter.update_pdf()
    return model


def main():
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    G = np.load(f"{HERE}/golden_forward.npz")
    torch.manual_seed(0)
    mods = GF.import_reference_model()
    make_differentiable_shims()
    sysm = import_reference_system()
    IA = mods["ia"]
    registry = mods["registry"]
    registry["none"] = type("Dummy", (nn.Module,), {"__init__": lambda self, c=None: nn.Module.__init__(self), "forward": lambda self, *a, **k: None})
    wrap, rd, _ = GF.build_rig(mods)
    registry["prebuilt"] = lambda cfg: wrap
    rays = torch.from_numpy(G["rays"])
    bg = torch.from_numpy(G["background_color"])
    rgb_t, alpha_t = targets(G)
    out = dict(target_rgb=rgb_t, target_alpha=alpha_t, subsample=np.int64(SUBSAMPLE))
    # the RngLog seed of golden_forward.npz's train run is 1000 + len(out) at that point of make_golden_forward.main; recover it by
    # matching the first stored draw instead of hard-coding the count
    def seed_of(tag):
        ref_near = G[f"{tag}_rng_1"]
        return next(s for s in range(1000, 1800) if np.array_equal(
            N(torch.rand((ref_near.shape[0],), generator=torch.Generator().manual_seed(s))), ref_near))
    seeds = {}
    for name, loss_cfg in VARIANTS.items():
        TAG, mode, spp = RUN_OF[name]
        seed = seeds.setdefault(TAG, seed_of(TAG))
        torch.manual_seed(0)
        with GF.RngLog(seed) as rng:
            GF.RNG = rng
            model = build_model(mods, IA, rd, bg, G["hdri"], mode, spp)
            with torch.no_grad():
                for c in model.material.network.lipshitz_bound_per_layer:
                    c.mul_(LIPSHITZ_SCALE[name])
            fake = _SystemStandIn(sysm, model, loss_cfg)
            batch = dict(rays=rays.clone(), rgb=torch.from_numpy(rgb_t), alpha=torch.from_numpy(alpha_t))
            loss = sysm.IntrinsicAvatarSystem.training_step(fake, batch, 0)["loss"]
            loss.backward()
        res = fake.out
        # same forward as the train run of golden_forward.npz, bit for bit
        for k in (G[TAG + "_out_keys"] if LIPSHITZ_SCALE[name] == 1.0 else ("comp_rgb", "comp_normal", "opacity", "num_samples", "ray_indices")):
            assert np.array_equal(N(res[str(k)]), G[f"{TAG}_out_{k}"]), ("forward differs from golden_forward.npz", k)
        out[f"{name}_run"] = np.array(TAG)
        out[f"{name}_loss"] = np.float64(loss.item())
        out[f"{name}_lipshitz_scale"] = np.float64(LIPSHITZ_SCALE[name])
        for k in ("comp_rgb_phys_full", "comp_albedo_full", "comp_roughness_full", "comp_metallic_full"):
            out[f"{name}_out_{k}"] = N(res[k])
        out[f"{name}_loss_terms"] = np.array(sorted(f"{k}={v!r}" for k, v in fake.logged.items() if k.startswith("train/")))
        out[f"{name}_lambdas"] = np.array(sorted(f"{k}={fake.C(v)!r}" for k, v in loss_cfg.items()))
        names = []
        for pname, p in model.named_parameters():
            if p.grad is None or p.numel() == 0:
                continue
            g = N(p.grad)
            assert np.isfinite(g).all(), pname
            if pname.endswith("encoding.encoding.params"):
                for k, v in table_gradient_summary(g.reshape(-1)).items():
                    out[f"{name}_grad_{pname}:{k}"] = v
            else:
                out[f"{name}_grad_{pname}"] = g
            names.append(pname)
        out[f"{name}_grad_names"] = np.array(names)
        print(name, "loss", loss.item(), {k: round(v, 8) for k, v in fake.logged.items() if k.startswith("train/loss")})
        print("   gradient groups:", {n_: float(np.abs(N(dict(model.named_parameters())[n_].grad)).max()) for n_ in names})
    torch.Tensor.cuda, torch.cuda.device = mods["restore"]
    np.savez_compressed(f"{HERE}/golden_backward.npz", **out)
    print("golden_backward.npz", os.path.getsize(f"{HERE}/golden_backward.npz") // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
