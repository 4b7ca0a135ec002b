#!/usr/bin/env python3
"""Golden vectors of the WHOLE hot path: the reference's own `IntrinsicAvatarModel.forward_`
(/root/reference/models/intrinsic_avatar.py:950-1651, imported by file path, never copied) run in the build container.

  python tests/golden/make_golden_forward.py          (needs /root/reference; CPU only; a few minutes)

What runs is the reference's unmodified Python:
    models/intrinsic_avatar.py        IntrinsicAvatarModel: setup, update_step, prepare_test_occupancy_grid, forward_,
                                      compute_indirect_radiance, pbr_{light,uniform_light,mis,mats}_forward
    models/rf/{geometry,radiance,density}.py, models/pbr/material.py, models/network_utils.py     the field modules
    models/deformers/{deformer,snarf_deformer,non_rigid_deformer}.py, fast_snarf/deformer_torch.py   the deformer
    models/volrend.py, models/pbr/utils.py, models/utils.py, lib/nerfacc/*.py                        as make_golden_host.py
underneath it, for what the reference tree has no CPU implementation of (tests-side shims, never part of the package):
    K1..K7, K8..K10   the reference's own kernel bodies compiled for the host (make_golden.py)
    pip nerfacc 0.5.3 the CPU oracle under the nerfacc names (make_golden_host.py)
    tinycudann        HashGrid / SphericalHarmonics in plain torch (tests/torch_ref.py: Instant-NGP definitions, differentiable)
    lib.torch_pbr     oracle/pbr_ref.py (numpy) under the class names (EnvironmentLightTensor, MultiLobe), explicit uniforms
    SMPL              absent (licence): the synthetic 24-bone rig of intrinsicavatar_amd/synthetic.py supplies tfs / w2s /
                      vertices and the skinning-weight grid; ForwardDeformer.switch_to_explicit / precompute / search run as is
Every random tensor the reference draws (torch.rand / rand_like / randn_like and the emitter / scatterer uniforms) is recorded
in call order and stored, so the oracle and the HIP path can be driven with the same numbers.

Only DATA is written (tests/golden/golden_forward.npz): the scene (MLP weights, rig, rays, occupancy grid; the two 50 MB
hash tables are a closed-form function of the entry index, synthetic.hash_table_values), the random tensors, and for every
run the output dict of forward_ (:1492-1651) plus the intermediate tensors the tests compare.
"""
import importlib.util
import math
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

from intrinsicavatar_amd import synthetic as S      # noqa: E402  (numpy-only helpers: rig, rays, formula-generated tables)

HASH = dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.447269237440378)
RES = 32                      # ForwardDeformer resolution: skinning grid [1,24,8,32,32]
HW = 28                       # frame: HW x HW rays
N = lambda t: t.detach().cpu().numpy()      # noqa: E731
_TORCH_RAND = torch.rand
FORMULA_MIN = 100_000          # random tensors with more elements are a closed-form function of the index (not stored)


class Cfg(dict):
    """attribute-style config (what the reference reads through OmegaConf: .key, ['key'], .get, `in`, .copy())."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


# ----------------------------------------------------------------------------- explicit RNG: record every draw
class RngLog:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.log = []
        self.orig = (torch.rand, torch.rand_like, torch.randn_like)

    def rand(self, *size, **kw):
        size = tuple(size[0] if (len(size) == 1 and not isinstance(size[0], int)) else size)
        if int(np.prod(size)) >= FORMULA_MIN:                          # the [n_rays, 512] shuffle uniforms of uniform_light
            seed = 7000 + len(self.log)
            t = torch.from_numpy((S.hash_table_values(int(np.prod(size)), seed, 1.0).astype(np.float64) + 1.0) / 2.0).float().reshape(size)
            self.log.append(("rand_formula", torch.tensor([seed, int(np.prod(size))], dtype=torch.int64)))
            return t
        t = self.orig[0](size, generator=self.g)
        self.log.append(("rand", t.clone()))
        return t

    def rand_like(self, x, **kw):
        dt = kw.get("dtype", x.dtype if x.dtype.is_floating_point else torch.float32)
        if x.numel() >= FORMULA_MIN:        # the 3 x 64^3 voxel jitter of the occupancy grid: u[i] = (hash_table_values(n, seed, 1)[i] + 1) / 2
            seed = 7000 + len(self.log)
            t = torch.from_numpy((S.hash_table_values(x.numel(), seed, 1.0).astype(np.float64) + 1.0) / 2.0).to(dt).reshape(x.shape)
            self.log.append(("rand_like_formula", torch.tensor([seed, x.numel()], dtype=torch.int64)))
            return t
        t = self.orig[0](tuple(x.shape), generator=self.g).to(dt)
        self.log.append(("rand_like", t.clone()))
        return t

    def randn_like(self, x, **kw):
        t = torch.randn(tuple(x.shape), generator=self.g)
        self.log.append(("randn_like", t.clone()))
        return t

    def uniforms(self, tag, shape):
        t = self.orig[0](tuple(shape), generator=self.g)
        self.log.append((tag, t.clone()))
        return t

    def __enter__(self):
        torch.rand, torch.rand_like, torch.randn_like = self.rand, self.rand_like, self.randn_like
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn_like = self.orig


RNG = None      # the active RngLog (the torch_pbr shim draws its uniforms from it)


# ----------------------------------------------------------------------------- tinycudann on CPU (plain torch)
def make_tcnn_shim():
    from tests import torch_ref as TR
    m = types.ModuleType("tinycudann")

    class Encoding(nn.Module):
        def __init__(self, n_input_dims, encoding_config, dtype=torch.float32):
            super().__init__()
            self.n_input_dims = n_input_dims
            self.otype = encoding_config["otype"]
            if self.otype == "HashGrid":
                c = encoding_config
                self.cfg = TR.hash_cfg(c["n_levels"], c["log2_hashmap_size"], c["base_resolution"], c["per_level_scale"])
                self.F = c["n_features_per_level"]
                assert self.F == 2 and n_input_dims == 3
                self.n_output_dims = c["n_levels"] * self.F
                self.params = nn.Parameter(torch.zeros(self.cfg[0][-1] * self.F))
            elif self.otype == "SphericalHarmonics":
                assert encoding_config["degree"] == 4 and n_input_dims == 3
                self.n_output_dims = 16
                self.params = nn.Parameter(torch.zeros(0))
            else:
                raise NotImplementedError(self.otype)

        def forward(self, x):
            if self.otype == "HashGrid":
                return TR.hashgrid(x, self.params.view(-1, 2), self.cfg)
            return TR.sh4(x)
    m.Encoding = Encoding
    m.free_temporary_memory = lambda: None
    sys.modules["tinycudann"] = m
    return m


# ----------------------------------------------------------------------------- lib.torch_pbr on CPU (oracle/pbr_ref.py)
def make_torch_pbr_shim():
    from oracle import pbr_ref as Pb
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))      # noqa: E731
    m = types.ModuleType("lib.torch_pbr")

    class EnvironmentLightTensor(nn.Module):
        def __init__(self, config):
            super().__init__()
            ec = config["envlight_config"]
            g = torch.Generator().manual_seed(0)
            res = ec["base_res"]
            self.base = nn.Parameter(_TORCH_RAND((res, 2 * res, 3), generator=g) * ec["scale"] + ec["bias"])
            self.pdf_scale = None
            self._pmf = None

        def update_pdf(self):
            self._pmf = Pb.envlight_pmf(N(self.base))

        def sample(self, k):
            u = N(RNG.uniforms("emitter.sample", (k, 3))).astype(np.float64)
            return T(Pb.envlight_sample(self._pmf, k, u[:, 0], u[:, 1], u[:, 2]))

        def pdf(self, d):
            return T(Pb.envlight_pdf(self._pmf, N(d).astype(np.float32)))[:, None]

        def eval(self, d):
            return T(Pb.envlight_eval(N(self.base), N(d).astype(np.float32)))

        def sample_uniform_sphere_stratified(self, n_rays, n_theta, n_phi, device=None):
            u = N(RNG.uniforms("emitter.sample_uniform_sphere_stratified", (n_theta * n_phi, 2)))
            d, ip = Pb.uniform_sphere_stratified(n_theta, n_phi, u)
            return T(d), T(ip)

        def generate_image(self):
            return self.base

    class MultiLobe(nn.Module):
        def __init__(self, config=None):
            super().__init__()

        def eval(self, wi, n, wo, alpha_x, alpha_y, albedo, metallic, attenuation):
            d, s = Pb.brdf_eval(N(n), N(wi), N(wo), N(alpha_x), N(albedo), N(metallic)[:, 0])
            return T(d), T(s)

        def sample(self, wi, n, alpha_x, alpha_y, albedo=None, metallic=None, attenuation=None):
            u = N(RNG.uniforms("scatterer.sample", (n.shape[0], 3)))
            return T(Pb.brdf_sample(N(n), N(wi), N(alpha_x), u))

        def pdf(self, wi, n, wo, alpha_x, alpha_y, albedo=None, metallic=None, attenuation=None):
            return T(Pb.brdf_pdf(N(n), N(wi), N(wo), N(alpha_x)))[:, None]

    def rgb_to_srgb(f):
        f = f.clamp(0.0, 1.0)
        return torch.where(f <= 0.0031308, f * 12.92, torch.pow(f.clamp_min(0.0031308), 1.0 / 2.4) * 1.055 - 0.055)

    m.EnvironmentLightTensor, m.MultiLobe, m.rgb_to_srgb = EnvironmentLightTensor, MultiLobe, rgb_to_srgb
    m.luminance = lambda rgb: 0.2126 * rgb[..., 0:1] + 0.7152 * rgb[..., 1:2] + 0.0722 * rgb[..., 2:3]
    for name in ("EnvironmentLightSG", "EnvironmentLightMLP", "EnvironmentLightNGP", "Mirror", "Lambertian", "GGX", "DiffuseSGGX",
                 "SpecularSGGX", "MultiLobeSGGX"):
        setattr(m, name, type(name, (nn.Module,), {}))
    sys.modules["lib.torch_pbr"] = m
    return m


# ----------------------------------------------------------------------------- the reference model on CPU
def import_reference_model():
    import make_golden as MG
    import make_golden_host as MH
    import contextlib
    make_tcnn_shim()
    mods = MH.import_reference_host_code()                    # nerfacc names (oracle), lib.nerfacc (host K1-K7), volrend, pbr.utils, occ
    make_torch_pbr_shim()
    _, snarf_host = MG.build_reference_host_modules()          # cached build of the K8-K10 bodies
    sys.modules["utils.misc"].get_rank = lambda: "cpu"
    sys.modules["systems.utils"].update_module_step = lambda m, e, s: m.update_step(e, s) if hasattr(m, "update_step") else None
    for name in ("lib.pytorch3d", "lib.pytorch3d.ops", "torchgeometry", "torchgeometry.core", "torchgeometry.core.conversions"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["lib.pytorch3d"].ops = sys.modules["lib.pytorch3d.ops"]
    sys.modules["torchgeometry.core"].conversions = sys.modules["torchgeometry.core.conversions"]
    # the occupancy-grid estimator of pip nerfacc, as far as the reference touches it (prepare_test_occupancy_grid :360-381;
    # its .sampling is replaced by the reference's own sampling_override at import, :144)
    nf = sys.modules["nerfacc"]

    class OccGridEstimator(nn.Module):
        def __init__(self, roi_aabb, resolution=64, levels=1):
            super().__init__()
            self.register_buffer("aabbs", torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(1, 6))
            self.register_buffer("binaries", torch.zeros((levels, resolution, resolution, resolution), dtype=torch.bool))
    nf.OccGridEstimator = OccGridEstimator
    M = sys.modules["models"]
    registry = {}
    M.models = registry
    M.register = lambda name: (lambda cls: registry.__setitem__(name, cls) or cls)
    M.make = lambda name, config: registry[name](config)
    MH.pkg("models.deformers", f"{REF}/models/deformers")
    MH.pkg("models.deformers.fast_snarf", f"{REF}/models/deformers/fast_snarf")
    smplx = types.ModuleType("models.deformers.smplx")
    smplx.SMPL = object
    sys.modules["models.deformers.smplx"] = smplx
    import torch.utils.cpp_extension as CE
    orig_load, orig_cuda, orig_dev = CE.load, torch.Tensor.cuda, torch.cuda.device
    CE.load = lambda name, **kw: snarf_host                       # deformer_torch.py:9-18 JIT-loads its three CUDA modules
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    try:
        MH.load("models.base", f"{REF}/models/base.py")
        MH.load("models.network_utils", f"{REF}/models/network_utils.py")
        for f in ("geometry", "radiance", "density"):
            MH.load(f"models.rf.{f}", f"{REF}/models/rf/{f}.py")
        MH.load("models.pbr.material", f"{REF}/models/pbr/material.py")
        mods["fd"] = MH.load("models.deformers.fast_snarf.deformer_torch", f"{REF}/models/deformers/fast_snarf/deformer_torch.py")
        mods["sd"] = MH.load("models.deformers.snarf_deformer", f"{REF}/models/deformers/snarf_deformer.py")
        mods["nrd"] = MH.load("models.deformers.non_rigid_deformer", f"{REF}/models/deformers/non_rigid_deformer.py")
        mods["dfm"] = MH.load("models.deformers.deformer", f"{REF}/models/deformers/deformer.py")
        mods["ia"] = MH.load("models.intrinsic_avatar", f"{REF}/models/intrinsic_avatar.py")
    finally:
        CE.load = orig_load
    mods["restore"] = (orig_cuda, orig_dev)
    mods["registry"] = registry
    tp = sys.modules["lib.torch_pbr"]
    registry["envlight-tensor"], registry["brdf-multi-lobe"] = tp.EnvironmentLightTensor, tp.MultiLobe     # models/__init__.py:39-51
    return mods


def model_config(render_mode, spp, gi):
    enc = Cfg(otype="ProgressiveBandHashGrid", include_xyz=True, start_level=4, update_steps=125, start_step=500, interpolation="Linear", **HASH)
    return Cfg(
        name="intrinsic-avatar", global_illumination=gi, render_mode=render_mode, scene_aabb=[-1.25, -1.55, -1.25, 1.25, 0.95, 1.25],
        samples_per_pixel=spp, num_samples_per_ray=64, num_samples_per_secondary_ray=64, secondary_shader_chunk=160000,
        secondary_near_plane=0.0, secondary_far_plane=1.5, secondary_importance_sample=True, zero_crossing_search=True,
        resample_light=True, volume_scattering=True, add_emitter=False, grid_prune=True, grid_prune_occ_thre=0.001,
        grid_prune_ema_decay=0.8, randomized=True, ray_chunk=4096, learned_background=False, learn_material=True,
        material_feature="hybrid", phys_kick_in_step=10000, importance_sample_kick_in_step=1000, background_color="random",
        density=Cfg(name="learned-laplace-density", beta_schedule_steps=10000, params_init=Cfg(beta=0.05)),
        pose_encoder=Cfg(name="none"), pose_correction=Cfg(name="none"), deformer=Cfg(name="prebuilt"),
        geometry=Cfg(name="volume-sdf", radius=1.0, feature_dim=13, isosurface=None, grad_type="analytic", finite_difference_eps="progressive",
                     xyz_encoding_config=Cfg(enc),
                     mlp_network_config=Cfg(otype="VanillaMLP", output_activation="none", n_neurons=64, n_hidden_layers=1,
                                            sphere_init=True, sphere_init_radius=0.5, weight_norm=True)),
        radiance=Cfg(name="volume-ref-dir-radiance", input_feature_dim=16, xyz_encoding_config=Cfg(enc),
                     dir_encoding_config=Cfg(otype="SphericalHarmonics", degree=4),
                     mlp_network_config=Cfg(otype="VanillaMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                            n_hidden_layers=2), color_activation="sigmoid"),
        material=Cfg(name="volume-material", input_feature_dim=48, n_output_dim=5, albedo_scale=0.77, albedo_bias=0.03,
                     roughness_scale=0.9, roughness_bias=0.09, metallic_scale=1.0, metallic_bias=0.0,
                     mlp_network_config=Cfg(otype="LipshitzMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                            n_hidden_layers=2), material_activation="sigmoid"),
        scatterer=Cfg(name="brdf-multi-lobe"),
        light=Cfg(name="envlight-tensor", xyz2lonlat_mode=None, envlight_config=Cfg(hdr_filepath=None, scale=0.5, bias=0.25, base_res=16)))


def build_rig(mods):
    """the reference's SNARFDeformer (rigid) + wrapper on the synthetic 24-bone rig: ForwardDeformer.switch_to_explicit builds
    offset / scale / bbox / the query_weights closure itself, precompute runs the reference's K10."""
    FD, SD, DFM, NRD = mods["fd"], mods["sd"], mods["dfm"], mods["nrd"]
    D, H, W = RES // 4, RES, RES
    wgrid, offk, sck, bbox = S.skinning_weight_grid(D, H, W, smooth_iters=5)
    rig = S.make_rig(S.make_pose(0))
    lo, hi = (S.JOINTS - S.RADII[:, None]).min(0), (S.JOINTS + S.RADII[:, None]).max(0)
    rd = SD.SNARFDeformer.__new__(SD.SNARFDeformer)
    rd.opt = Cfg(use_j_inv=False, optimize_betas=False, resolution=RES)
    fd = FD.ForwardDeformer(Cfg(version=1))
    fd.device = "cpu"
    fd.query_weights = lambda x, cond, mask=None: torch.from_numpy(wgrid)        # consumed once by switch_to_explicit
    fd.switch_to_explicit(resolution=RES, smpl_verts=torch.from_numpy(np.stack([lo, hi]).astype(np.float32))[None], smpl_weights=None,
                          use_smpl=False)
    assert np.allclose(N(fd.offset_kernel).reshape(3), offk.reshape(3), atol=1e-6) and np.allclose(N(fd.scale_kernel).reshape(3), sck.reshape(3), rtol=1e-6)
    rd.deformer = fd
    rd.dtype = torch.float32
    rd.initialized = True
    # canonical "vertices": the corner points of the bones' capsules (get_bbox_from_smpl only takes min / max)
    cano = np.concatenate([S.JOINTS - S.RADII[:, None], S.JOINTS + S.RADII[:, None]]).astype(np.float32)
    rd.bbox = SD.get_bbox_from_smpl(torch.from_numpy(cano)[None])
    tfs = torch.from_numpy(rig["tfs"])
    rd.tfs = tfs
    fd.precompute(tfs)
    rd.w2s = torch.from_numpy(rig["w2s"])[None]
    jp = rig["joints_posed"]
    rd.vertices = torch.from_numpy(np.concatenate([jp - S.RADII[:, None], jp + S.RADII[:, None]]).astype(np.float32))[None]
    rd.smpl_outputs = types.SimpleNamespace(betas=torch.zeros((1, 10)))
    rd.rot_mats, rd.basic_joints = None, None
    wrap = DFM.SNARFDeformer.__new__(DFM.SNARFDeformer)
    nn.Module.__init__(wrap)
    wrap.config, wrap.rank = Cfg(), "cpu"
    wrap.n_input_dims = wrap.n_output_dims = 3
    object.__setattr__(wrap, "rigid_deformer", rd)
    wrap.non_rigid_deformer = NRD.DummyNonRigidDeformer(Cfg())
    return wrap, rd, dict(lbs_voxel_final=N(fd.lbs_voxel_final), offset_kernel=N(fd.offset_kernel).reshape(3), scale_kernel=N(fd.scale_kernel).reshape(3),
                          cano_bbox=N(rd.bbox), ref_voxel_J=N(fd.voxel_J), tfs=rig["tfs"], w2s=rig["w2s"], vertices=N(rd.vertices))


def init_params(model, seed=0):
    """deterministic parameters: the hash tables are a closed-form function of the entry index (not stored), the MLPs keep the
    reference's own initialisers (seeded) plus the perturbations synthetic.build_frame applies."""
    with torch.no_grad():
        ge, re_ = model.geometry.encoding.encoding.encoding, model.radiance.xyz_encoding.encoding.encoding
        ge.params.copy_(torch.from_numpy(S.hash_table_values(ge.params.numel(), 11, 1e-2)))
        re_.params.copy_(torch.from_numpy(S.hash_table_values(re_.params.numel(), 12, 1e-2)))
        gw = torch.Generator().manual_seed(seed + 3)
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = _TORCH_RAND((64, 32), generator=gw) * 0.04 - 0.02


def hdri(Hh=16, Ww=32):
    v, u = np.meshgrid((np.arange(Hh) + 0.5) / Hh, (np.arange(Ww) + 0.5) / Ww, indexing="ij")
    sky = np.stack([0.3 + 0.4 * (1 - v), 0.4 + 0.4 * (1 - v), 0.6 + 0.4 * (1 - v)], -1)
    img = np.where((v < 0.5)[..., None], sky, np.full((Hh, Ww, 3), 0.08))
    sun = 40.0 * np.exp(-(((u - 0.3) * 2) ** 2 + ((v - 0.25) * 2) ** 2) / (2 * 0.08 ** 2))
    return (img + sun[..., None]).astype(np.float32)


def main():
    global RNG
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    torch.manual_seed(0)
    mods = import_reference_model()
    IA = mods["ia"]
    out = {}
    registry = mods["registry"]
    dummy = type("Dummy", (nn.Module,), {"__init__": lambda self, c=None: nn.Module.__init__(self), "forward": lambda self, *a, **k: None})
    registry["none"] = dummy
    wrap, rd, rig_arrays = build_rig(mods)
    registry["prebuilt"] = lambda cfg: wrap
    out.update({"rig_" + k: v for k, v in rig_arrays.items()})
    rays = torch.from_numpy(S.camera_rays(HW, HW))
    out["rays"] = N(rays)
    bg = torch.tensor([0.2, 0.4, 0.6])
    out["background_color"] = N(bg)
    out["hdri"] = hdri()
    runs = [("light", 16, False, False), ("light", 64, True, False), ("uniform_light", 512, True, False), ("mis", 16, True, False),
            ("mats", 16, True, False), ("light", 16, True, True),
            ("uniform_light", 512, True, True)]                                  # (render_mode, spp, global_illumination, training)
    # the last run is the estimator the reference SHIPS for training (configs/config.yaml:46-48: render_mode uniform_light,
    # samples_per_pixel 512) in train() mode; runs are only ever appended: the RngLog seed of a run is 1000 + len(out) when it starts
    state_saved = False
    for mode, spp, gi, training in runs:
        tag = f"{mode}_{spp}_{'gi' if gi else 'nogi'}{'_train' if training else ''}"
        torch.manual_seed(0)                                                     # identical MLP initialisation in every run
        with RngLog(1000 + len(out)) as RNG:
            model = IA.IntrinsicAvatarModel(model_config(mode, spp, gi))
            init_params(model)
            model.eval()
            model.update_step(250, 25000)                                        # systems/base.py:150: all levels on, enable_phys, importance sampling
            model.train(training)
            assert model.enable_phys and model.importance_sample
            model.background_color = bg
            model.geometry.prepare_bbox(rd.bbox)
            model.radiance.prepare_bbox(rd.bbox)
            model.jitter_materials = training
            model.with_curvature_loss = False
            model.cond = None
            n0 = len(RNG.log)
            if training:
                # the training grid (TemporalOccGridEstimator, one level): filled by the reference's own update (:178-208)
                model.t_idx = 0.0
                model.occupancy_grid._update(step=0, t_idx=0, occ_eval_fn=lambda x: _occ_eval(model, x), occ_thre=0.001, ema_decay=0.8)
                binaries, aabb = model.occupancy_grid.binaries, model.occupancy_grid.aabbs
            else:
                model.prepare_test_occupancy_grid()
                binaries, aabb = model.occupancy_grid_test.binaries, model.occupancy_grid_test.aabbs
            model.emitter.base = nn.Parameter(torch.from_numpy(out["hdri"]))      # prepare(), :292-305
            model.emitter.pdf_scale = (model.emitter.base.shape[0] * model.emitter.base.shape[1]) / (2 * np.pi * np.pi)
            model.emitter.update_pdf()
            if not training:
                model.secondary_rays_d = model.emitter.sample(model.samples_per_pixel)
            with torch.set_grad_enabled(training):
                res = model.forward_(rays.clone())
            log = RNG.log
        if not state_saved:
            sd = model.state_dict()
            keep = [k for k in sd if not k.endswith("encoding.encoding.params") and not k.startswith("occupancy_grid") and sd[k].numel() > 0]
            out["state_keys"] = np.array(sorted(keep))
            out.update({"state_" + k: N(sd[k]) for k in keep})
            out["render_step_size"] = np.float64(model.render_step_size)
            state_saved = True
        out[tag + "_occ_binaries"], out[tag + "_occ_aabb"] = N(binaries), N(aabb)
        out[tag + "_rng_kinds"] = np.array([k for k, _ in log])
        for i, (_, t) in enumerate(log):
            out[f"{tag}_rng_{i}"] = N(t)
        out[tag + "_out_keys"] = np.array(sorted(res.keys()))
        for k, v in res.items():
            out[f"{tag}_out_{k}"] = N(v)
        print(tag, {k: tuple(v.shape) for k, v in res.items() if k in ("comp_rgb", "comp_rgb_phys", "num_samples")}, "rng draws:", [(k, tuple(t.shape)) for k, t in log],
              "n_samples", int(res["num_samples"][0]))
    torch.Tensor.cuda, torch.cuda.device = mods["restore"]
    np.savez_compressed(f"{HERE}/golden_forward.npz", **out)
    print("golden_forward.npz", os.path.getsize(f"{HERE}/golden_forward.npz") // 1024, "KiB,", len(out), "arrays")


def _occ_eval(model, x):
    """occ_eval_fn of IntrinsicAvatarModel.update_step (:186-198)."""
    def geometry_fn(p):
        return model.geometry(p, with_grad=False, with_feature=False, with_laplace=False)
    _, sdf, *_ = model.deformer(x, model.cond, geometry_fn, with_jac=False, eval_mode=True)
    density = model.density(sdf)
    return 1.0 - torch.exp(-density * model.render_step_size)


if __name__ == "__main__":
    main()
