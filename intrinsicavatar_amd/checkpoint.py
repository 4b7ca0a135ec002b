"""Checkpoint key layout of the modules on the path.

A reference checkpoint is a Lightning file whose `state_dict` holds the system's parameters under `model.<component>.`
(launch.py:110-125 loads it with strict=False after dropping occupancy-grid and pose-correction entries).  The field
modules of this package register their parameters under the SAME names as the reference classes
(tests/golden/golden_state_keys.json, produced from the reference's classes by tests/golden/make_golden_keys.py), so a
component's slice of a reference state_dict loads with strict=True and vice versa.

Layouts inside the tensors: MLP weights are torch [out, in] matrices in the reference's input order (the kernels' column
order is produced on the fly by `effective_weights`); the hash table is tiny-cuda-nn's flat `params` vector -- level-major,
then entry, then feature (the Instant-NGP layout; tiny-cuda-nn itself is not in /root/reference: unpinned)."""
from typing import Dict, Optional

import torch

COMPONENTS = ("geometry", "radiance", "density", "material")
_DROPPED = ("occupancy_grid", "pose_correction")          # launch.py:113-123 (test mode)


def split_reference_state_dict(state_dict: Dict[str, torch.Tensor], prefix: str = "model.") -> Dict[str, Dict[str, torch.Tensor]]:
    """{component: {key without 'model.<component>.': tensor}}; entries launch.py drops at test time are dropped here too."""
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in state_dict.items():
        if not k.startswith(prefix) or any(d in k for d in _DROPPED):
            continue
        comp, _, rest = k[len(prefix):].partition(".")
        out.setdefault(comp, {})[rest] = v
    return out


def load_reference_state_dict(rs, state_dict: Dict[str, torch.Tensor], material=None, prefix: str = "model.",
                              strict: bool = True) -> Dict[str, list]:
    """load geometry / radiance / density (/ material) of a RenderStep from a reference `state_dict`.
    returns {component: [keys of the checkpoint that belong to components this package does not model]}."""
    parts = split_reference_state_dict(state_dict, prefix)
    mods = dict(geometry=rs.geometry, radiance=rs.radiance, density=rs.density)
    if material is not None:
        mods["material"] = material
    for name, mod in mods.items():
        if name not in parts:
            if strict:
                raise KeyError(f"checkpoint has no '{prefix}{name}.*' entries")
            continue
        mod.load_state_dict(parts[name], strict=strict)
    return {k: sorted(v) for k, v in parts.items() if k not in mods}


def reference_state_dict(rs, material=None, prefix: str = "model.") -> Dict[str, torch.Tensor]:
    """the inverse: this package's parameters under the reference's checkpoint keys."""
    out = {}
    mods = dict(geometry=rs.geometry, radiance=rs.radiance, density=rs.density)
    if material is not None:
        mods["material"] = material
    for name, mod in mods.items():
        for k, v in mod.state_dict().items():
            out[f"{prefix}{name}.{k}"] = v
    return out
