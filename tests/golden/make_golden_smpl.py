"""Golden vectors for the SMPL forward kinematics on the caller side of the path.

Runs ONLY in the build container: imports the reference's own models/deformers/smplx/lbs.py from /root/reference (pure
PyTorch) on a small synthetic body model (the real SMPL pkl is not available) and stores inputs + outputs.
    python tests/golden/make_golden_smpl.py      ->  tests/golden/golden_smpl.npz
Only data is committed; nothing of the reference's source travels."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/models/deformers/smplx"
HERE = os.path.dirname(os.path.abspath(__file__))

PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]      # SMPL kinematic tree


def load_ref_lbs():
    pkg = types.ModuleType("refsmplx")
    pkg.__path__ = [REF]
    sys.modules["refsmplx"] = pkg
    for name in ("utils", "lbs"):
        spec = importlib.util.spec_from_file_location(f"refsmplx.{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refsmplx.{name}"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["refsmplx.lbs"]


def synthetic_body(g, V=64, NB=10, J=24):
    v_template = torch.randn(V, 3, generator=g, dtype=torch.float64) * 0.4
    shapedirs = torch.randn(V, 3, NB, generator=g, dtype=torch.float64) * 0.02
    posedirs = torch.randn((J - 1) * 9, V * 3, generator=g, dtype=torch.float64) * 0.005
    J_regressor = torch.rand(J, V, generator=g, dtype=torch.float64) ** 8
    J_regressor = J_regressor / J_regressor.sum(1, keepdim=True)
    lbs_weights = torch.softmax(torch.randn(V, J, generator=g, dtype=torch.float64) * 3.0, -1)
    return v_template, shapedirs, posedirs, J_regressor, lbs_weights


def main():
    lbs = load_ref_lbs()
    g = torch.Generator().manual_seed(7)
    v_template, shapedirs, posedirs, J_regressor, lbs_weights = synthetic_body(g)
    parents = torch.tensor(PARENTS, dtype=torch.long)
    B = 5
    betas = torch.randn(B, 10, generator=g, dtype=torch.float64)
    pose = torch.randn(B, 72, generator=g, dtype=torch.float64) * 0.4
    pose[1] = 0.0                                  # rest pose (exercises the +1e-8 in the axis-angle norm)
    pose[2, 3:] = 0.0                              # global orientation only
    transl = torch.randn(B, 3, generator=g, dtype=torch.float64)
    out = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        c = lambda t: t.to(dt)      # noqa: E731
        verts, joints, A, T, shape_off, pose_off = lbs.lbs(c(betas), c(pose), c(v_template), c(shapedirs), c(posedirs),
                                                           c(J_regressor), parents, c(lbs_weights))
        out[f"verts_{tag}"], out[f"joints_{tag}"], out[f"A_{tag}"] = verts.numpy(), joints.numpy(), A.numpy()
        out[f"rodrigues_{tag}"] = lbs.batch_rodrigues(c(pose).view(-1, 3)).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_smpl.npz"), v_template=v_template.numpy(), shapedirs=shapedirs.numpy(),
                        posedirs=posedirs.numpy(), J_regressor=J_regressor.numpy(), lbs_weights=lbs_weights.numpy(),
                        parents=np.array(PARENTS), betas=betas.numpy(), pose=pose.numpy(), transl=transl.numpy(), **out)
    print("wrote golden_smpl.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
