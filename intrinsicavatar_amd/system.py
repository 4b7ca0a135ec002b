"""The two callers either side of render_step ("adjacent step either side of the path"), host side:

    preprocess_data   systems/intrinsic_avatar.py:84-158    a dataset batch -> what the model is called with: rays [n,8] =
                                                            (o, d, near, far), the training targets composited over the step's
                                                            background colour (sRGB-encoded, as the images are), flat masks / maps
    chunk_batch       models/utils.py:16-61                 run a function over slices of its batched tensor arguments, merge the results
    model_forward     models/intrinsic_avatar.py:1653-1666  IntrinsicAvatarModel.forward: forward_ as it is in training, chunk_batch over
                                                            `ray_chunk` rays with the results moved to the host in evaluation, + "beta"

Plain tensor plumbing on whatever device the batch lives on (no kernels: a few element-wise operators per frame); held to the reference's
own functions by tests/golden/golden_system.npz (tests/golden/make_golden_system.py)."""
from collections import defaultdict
from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

from . import pbr

BACKGROUNDS = ("white", "black", "random")


def background_for(mode: str, stage: str, device, generator: Optional[torch.Generator] = None) -> Tensor:
    """the step's background colour (:121-143): the configured one while training, white otherwise."""
    if stage not in ("train",):
        return torch.ones(3, dtype=torch.float32, device=device)
    if mode == "white":
        return torch.ones(3, dtype=torch.float32, device=device)
    if mode == "black":
        return torch.zeros(3, dtype=torch.float32, device=device)
    if mode == "random":
        return torch.rand(3, dtype=torch.float32, device=device, generator=generator)
    raise NotImplementedError


def preprocess_data(batch: Dict[str, Tensor], stage: str, background_color: str = "white", device=None,
                    generator: Optional[torch.Generator] = None, background: Optional[Tensor] = None) -> Tuple[Dict[str, Tensor], Tensor, float]:
    """in place on `batch`, like the reference; -> (batch, background colour [3], t_idx).  `background`: an explicit colour instead of
    the draw of the "random" mode (parity with a recorded run)."""
    flat3 = lambda k: batch[k].reshape(-1, 3)      # noqa: E731
    if "hdri" in batch:
        assert stage in ["test"]
        assert batch["hdri"].shape[0] == 1
        batch["hdri"] = batch["hdri"].squeeze(0)
    rays = torch.cat([batch.pop("rays_o"), batch.pop("rays_d"), batch.pop("near")[..., None], batch.pop("far")[..., None]], dim=-1).reshape(-1, 8)
    batch["rays"] = rays
    dev = device if device is not None else rays.device
    bg = background.to(dev).float() if background is not None else background_for(background_color, stage, dev, generator)
    if "rgb" in batch:
        rgb = flat3("rgb")
        fg = batch["alpha"].reshape(-1)
        batch["rgb_wo_mask"] = rgb
        batch["rgb"] = rgb * fg[..., None] + pbr.rgb_to_srgb(bg.to(rgb.device) * (1 - fg[..., None]))
        batch["alpha"] = fg
    if "valid_mask" in batch:
        batch["valid_mask"] = batch["valid_mask"].reshape(-1)
    for k in ("albedo", "normal"):
        if k in batch:
            batch[k] = flat3(k)
    t_idx = batch["t_idx"] if stage in ["train"] else 0.0          # (not used in val / test / predict)
    return batch, bg, t_idx


def chunk_batch(func: Callable, chunk_size: int, move_to_cpu: bool, *args, **kwargs):
    """`func` on [i, i + chunk_size) of every positional tensor whose leading dimension equals that of the FIRST tensor argument; the
    chunks' results concatenated along dimension 0 -- a tensor, a tuple / list (type kept) or a dict of tensors; chunks that return None
    are left out, nothing returned at all gives None; results are detached when gradients are off and moved to the host on request."""
    B = next((a.shape[0] for a in args if isinstance(a, Tensor)), None)
    parts = defaultdict(list)
    kind, width = None, 0
    for i in range(0, B, chunk_size):
        r = func(*[a[i:i + chunk_size] if isinstance(a, Tensor) and a.shape[0] == B else a for a in args], **kwargs)
        if r is None:
            continue
        kind = type(r)
        if isinstance(r, Tensor):
            r = {0: r}
        elif isinstance(r, (tuple, list)):
            width = len(r)
            r = dict(enumerate(r))
        elif not isinstance(r, dict):
            raise TypeError(f"Return value of func must be in type [torch.Tensor, list, tuple, dict], get {type(r)}.")
        for k, v in r.items():
            v = v if torch.is_grad_enabled() else v.detach()
            parts[k].append(v.cpu() if move_to_cpu else v)
    if kind is None:
        return None
    merged = {k: torch.cat(v, dim=0) for k, v in parts.items()}
    if kind is Tensor:
        return merged[0]
    if kind in (tuple, list):
        return kind([merged[i] for i in range(width)])
    return merged


def model_forward(rs, rays: Tensor, material, emitter, spp: int, light_u: Tensor, shuffle_u: Optional[Tensor] = None, *, training: bool = False,
                  ray_chunk: int = 1 << 19, move_to_cpu: bool = True, **kw) -> Dict[str, Tensor]:
    """IntrinsicAvatarModel.forward (:1653-1666) over RenderStep: training -> forward_train_ on the whole batch; evaluation ->
    chunk_batch(forward_, ray_chunk, True, rays) -- every chunk's dict moved to the host and concatenated (per-ray maps along the rays,
    `num_samples*` as one entry per chunk, like the reference's) -- and the density's beta added.  The light directions (light_u [spp,3])
    are shared by the chunks; shuffle_u [n,spp] is sliced with the rays."""
    if training:
        out = rs.forward_train_(rays, material, emitter, spp, light_u, shuffle_u=shuffle_u, **kw)
    elif shuffle_u is None:
        out = chunk_batch(lambda r: rs.forward_(r, material, emitter, spp, light_u, None, **kw), ray_chunk, move_to_cpu, rays)
    else:
        out = chunk_batch(lambda r, su: rs.forward_(r, material, emitter, spp, light_u, su, **kw), ray_chunk, move_to_cpu, rays, shuffle_u)
    return {**out, "beta": rs.density.get_beta()}
