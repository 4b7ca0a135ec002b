"""Golden list of the reference's checkpoint keys for the modules on the path.

Runs ONLY in the build container: instantiates the reference's own VolumeSDF / VolumeRefDirRadiance / LaplaceDensity /
VolumeMaterial classes from /root/reference with their shipped YAML settings and stub third-party packages (tinycudann's
Encoding is replaced by a module holding a flat `params` vector -- the NAME of that entry is what is pinned, its length
is Instant-NGP's table size) and writes {state_dict key: shape} to tests/golden/golden_state_keys.json.
    python tests/golden/make_golden_keys.py"""
import importlib.util
import json
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class Cfg(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return dict.get(self, k, d)


def cfg(d):
    return Cfg({k: cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def ngp_table_size(c):
    n = 0
    for l in range(c["n_levels"]):
        res = int(__import__("math").ceil(c["base_resolution"] * c["per_level_scale"] ** l - 1.0)) + 1
        size = min(res ** 3, 2 ** c["log2_hashmap_size"])
        n += (size + 7) // 8 * 8
    return n * c["n_features_per_level"]


class StubEncoding(nn.Module):
    def __init__(self, n_input_dims, config, dtype=torch.float32):
        super().__init__()
        if config["otype"] == "HashGrid":
            self.params = nn.Parameter(torch.zeros(ngp_table_size(config)))
            self.n_output_dims = config["n_levels"] * config["n_features_per_level"]
        else:      # SphericalHarmonics: no parameters
            self.params = nn.Parameter(torch.zeros(0))
            self.n_output_dims = config["degree"] ** 2
        self.n_input_dims = n_input_dims


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    for name in ("tinycudann", "cv2", "pytorch_lightning", "pytorch_lightning.utilities", "nerfacc", "lib", "lib.torch_pbr",
                 "pytorch_lightning.utilities.rank_zero", "omegaconf", "systems", "systems.utils", "utils", "utils.misc"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["tinycudann"].Encoding = StubEncoding
    sys.modules["pytorch_lightning.utilities.rank_zero"].rank_zero_debug = lambda *a, **k: None
    sys.modules["pytorch_lightning.utilities.rank_zero"].rank_zero_info = lambda *a, **k: None
    sys.modules["systems.utils"].update_module_step = lambda *a, **k: None
    sys.modules["utils.misc"].config_to_primitive = lambda c: dict(c)
    sys.modules["utils.misc"].get_rank = lambda: "cpu"      # no GPU in the build container:
    import contextlib
    torch.cuda.device = lambda idx: contextlib.nullcontext()        # `with torch.cuda.device(get_rank())` becomes a no-op
    torch.Tensor.cuda = lambda self, *a, **k: self                   # `.cuda()` on constants (density.py:23) stays on the host
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["lib.torch_pbr"].luminance = lambda x: x
    models = types.ModuleType("models")
    models.models = {}

    def register(name):
        def deco(cls):
            models.models[name] = cls
            return cls
        return deco
    models.register = register
    sys.modules["models"] = models
    load("models.utils", f"{REF}/models/utils.py")
    load("models.base", f"{REF}/models/base.py")
    load("models.network_utils", f"{REF}/models/network_utils.py")
    sys.modules.setdefault("models.rf", types.ModuleType("models.rf"))
    sys.modules.setdefault("models.pbr", types.ModuleType("models.pbr"))
    load("models.rf.geometry", f"{REF}/models/rf/geometry.py")
    load("models.rf.radiance", f"{REF}/models/rf/radiance.py")
    load("models.rf.density", f"{REF}/models/rf/density.py")
    load("models.pbr.material", f"{REF}/models/pbr/material.py")
    grid = dict(otype="ProgressiveBandHashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                base_resolution=16, per_level_scale=1.447269237440378, interpolation="Linear", include_xyz=True,
                start_level=4, update_steps=125, start_step=500)
    geo_cfg = cfg(dict(name="volume-sdf", radius=1.0, feature_dim=13, isosurface=None, grad_type="analytic", finite_difference_eps="progressive",
                       xyz_encoding_config=grid,
                       mlp_network_config=dict(otype="VanillaMLP", output_activation="none", n_neurons=64, n_hidden_layers=1,
                                               sphere_init=True, sphere_init_radius=0.5, weight_norm=True)))
    rad_cfg = cfg(dict(name="volume-ref-dir-radiance", input_feature_dim=16, xyz_encoding_config=grid,
                       dir_encoding_config=dict(otype="SphericalHarmonics", degree=4),
                       mlp_network_config=dict(otype="VanillaMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                               n_hidden_layers=2), color_activation="sigmoid"))
    mat_cfg = cfg(dict(name="volume-material", input_feature_dim=48, n_output_dim=5, albedo_scale=0.77, albedo_bias=0.03,
                       roughness_scale=0.9, roughness_bias=0.09, metallic_scale=1.0, metallic_bias=0.0,
                       mlp_network_config=dict(otype="LipshitzMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                               n_hidden_layers=2), material_activation="sigmoid"))
    den_cfg = cfg(dict(name="learned-laplace-density", beta_schedule_steps=10000, params_init=dict(beta=0.3)))
    out = {}
    for comp, name, c in (("geometry", "volume-sdf", geo_cfg), ("radiance", "volume-ref-dir-radiance", rad_cfg),
                          ("density", "learned-laplace-density", den_cfg), ("material", "volume-material", mat_cfg)):
        m = models.models[name](c)
        out[comp] = {k: list(v.shape) for k, v in m.state_dict().items()}
        print(comp, out[comp])
    json.dump(out, open(os.path.join(HERE, "golden_state_keys.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
