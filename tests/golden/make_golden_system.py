#!/usr/bin/env python3
"""Golden vectors for the two callers either side of render_step, made by running the REFERENCE's own code:

    IntrinsicAvatarSystem.preprocess_data   /root/reference/systems/intrinsic_avatar.py:84-158   (imported by file path, called unbound
                                            with a stand-in `self`: config.model.background_color, rank, a model whose prepare() records)
    chunk_batch                             /root/reference/models/utils.py:16-61                (what IntrinsicAvatarModel.forward
                                            :1653-1666 wraps forward_ in for evaluation)

  python tests/golden/make_golden_system.py          (build container only: needs /root/reference; CPU, ~1 min)

Only DATA is written (tests/golden/golden_system.npz): the input batches, the output batches, and for chunk_batch the results of four
small closures that the test defines identically (one per return type the function distinguishes).  `rgb_to_srgb` comes from
`lib.torch_pbr`, an empty submodule of the reference: the stand-in is make_golden_forward.py's (the standard sRGB transfer function)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

import make_golden_forward as GF      # noqa: E402
import make_golden_backward as GB     # noqa: E402

N = GF.N
Cfg = GF.Cfg


def closures():
    """the functions chunk_batch is run on -- tests/test_system_cpu.py defines the same four."""
    def f_tensor(x, y, scale=1.0):
        return (x * scale + y.sum(-1, keepdim=True)).float()

    def f_tuple(x, y, scale=1.0):
        return x * scale, y[:, :2] - 1.0

    def f_list(x, y, scale=1.0):
        return [x.sum(-1), (y * scale).cumsum(-1)]

    def f_dict(x, y, scale=1.0):
        if x.shape[0] and float(x[0, 0]) < -0.5:          # a chunk that returns nothing is skipped (:34-35)
            return None
        return dict(a=x * scale, b=y.mean(-1), n=torch.full((x.shape[0],), x.shape[0], dtype=torch.int32))
    return dict(tensor=f_tensor, tuple=f_tuple, list=f_list, dict=f_dict)


class _Model:
    def __init__(self):
        self.background_color = None
        self.t_idx = None
        self.prepared = []

    def prepare(self, batch):
        self.prepared.append(sorted(batch.keys()))


class _Self:
    def __init__(self, mode, rank="cpu"):
        self.config = Cfg(model=Cfg(background_color=mode))
        self.rank = rank
        self.model = _Model()
        self.train_num_rays = None


def main():
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    mods = GF.import_reference_model()
    tp = sys.modules["lib.torch_pbr"]              # (the two names systems/intrinsic_avatar.py:13 imports besides rgb_to_srgb; not called here)
    tp.luma = lambda x: ((x[..., 0:1] + x[..., 1:2] + x[..., 2:3]) / 3.0).expand_as(x)
    tp.max_value = lambda x: torch.max(x, dim=-1, keepdim=True)[0].expand_as(x)
    sysm = GB.import_reference_system()
    utils = sys.modules["models.utils"]
    out = {}
    g = torch.Generator().manual_seed(11)
    # ---- preprocess_data: a [2, 6] "image" of rays in train stage with the three background modes, a test-stage batch with an HDRI
    H, W = 2, 6
    base = dict(rays_o=torch.randn((H, W, 3), generator=g), rays_d=torch.nn.functional.normalize(torch.randn((H, W, 3), generator=g), dim=-1),
                near=torch.rand((H, W), generator=g) + 3.0, far=torch.rand((H, W), generator=g) + 5.0,
                rgb=torch.rand((H, W, 3), generator=g), alpha=(torch.rand((H, W), generator=g) > 0.4).float() * torch.rand((H, W), generator=g),
                valid_mask=torch.rand((H, W), generator=g) > 0.2, albedo=torch.rand((H, W, 3), generator=g),
                normal=torch.randn((H, W, 3), generator=g), t_idx=torch.tensor([3]))
    for k, v in base.items():
        out[f"pre_in_{k}"] = N(v)
    cases = [("white", "train"), ("black", "train"), ("random", "train"), ("white", "test"), ("black", "validation")]
    out["pre_cases"] = np.array([f"{m}:{s}" for m, s in cases])
    for mode, stage in cases:
        fake = _Self(mode)
        batch = {k: v.clone() for k, v in base.items()}
        if stage == "test":
            batch["hdri"] = torch.rand((1, 4, 8, 3), generator=g)
            out["pre_in_hdri"] = N(batch["hdri"])
        torch.manual_seed(1234)                                         # the "random" background draws from the global generator
        sysm.IntrinsicAvatarSystem.preprocess_data(fake, batch, stage)
        tag = f"pre_{mode}_{stage}"
        out[f"{tag}_keys"] = np.array(sorted(batch.keys()))
        for k, v in batch.items():
            out[f"{tag}_out_{k}"] = N(v)
        out[f"{tag}_background_color"] = N(fake.model.background_color)
        out[f"{tag}_t_idx"] = np.float64(float(torch.as_tensor(fake.model.t_idx).reshape(-1)[0]))
        out[f"{tag}_train_num_rays"] = np.int64(fake.train_num_rays)
        out[f"{tag}_prepare_saw"] = np.array(fake.model.prepared[0])
    # ---- chunk_batch
    x, y = torch.randn((11, 3), generator=g), torch.randn((11, 4), generator=g)
    x[8, 0] = -1.0                                                       # the chunk starting at row 8 returns None from f_dict
    out["cb_x"], out["cb_y"] = N(x), N(y)
    for name, fn in closures().items():
        for chunk, to_cpu in ((4, False), (4, True), (11, False), (64, True)):
            r = utils.chunk_batch(fn, chunk, to_cpu, x, y, scale=0.5)
            tag = f"cb_{name}_{chunk}_{int(to_cpu)}"
            if isinstance(r, dict):
                out[f"{tag}_keys"] = np.array(sorted(r.keys()))
                for k, v in r.items():
                    out[f"{tag}_{k}"] = N(v)
            elif isinstance(r, (tuple, list)):
                out[f"{tag}_type"] = np.array(type(r).__name__)
                for i, v in enumerate(r):
                    out[f"{tag}_{i}"] = N(v)
            else:
                out[f"{tag}_0"] = N(r)
    assert utils.chunk_batch(lambda x, y, scale=1.0: None, 4, False, x, y) is None        # nothing returned at all (:54-55)
    np.savez_compressed(f"{HERE}/golden_system.npz", **out)
    print(len(out), "arrays ->", f"{HERE}/golden_system.npz", os.path.getsize(f"{HERE}/golden_system.npz"), "bytes")


if __name__ == "__main__":
    main()
