"""TemporalOccGridEstimator (models/occ_grid/temporal_occ_grid.py) and the per-frame test-time grid of
IntrinsicAvatarModel.prepare_test_occupancy_grid (models/intrinsic_avatar.py:307-381) on the MI355X kernels.

Buffers keep the reference's names (`resolution`, `aabbs`, `occs`, `binaries`: checkpoint-compatible).
Random numbers are explicit inputs."""
import math
from typing import Callable, Optional

import torch
from torch import Tensor

from . import _lib as L
from . import nerfacc


def binarize(occs: Tensor, resolution, thre_max: float, keep_largest_component: bool = True):
    """max_pool3d(3) -> thre = clamp(mean(>=0), max=thre_max) -> (> thre) -> largest connected component.
    occs: [rx*ry*rz] float. returns (binaries [rx,ry,rz] bool, thre [1])."""
    rx, ry, rz = (int(v) for v in resolution)
    occs = occs.contiguous().float()
    dev = occs.device
    binaries = torch.empty((rx, ry, rz), dtype=torch.bool, device=dev)
    thre = torch.empty(1, device=dev)
    tmp = torch.empty(int(L.lib().ia_occgrid_tmp_bytes(L.i32(rx), L.i32(ry), L.i32(rz))), dtype=torch.uint8, device=dev)
    L.check(L.lib().ia_occgrid_binarize(L.i32(rx), L.i32(ry), L.i32(rz), L.ptr(occs), L.f32(thre_max),
                                        L.i32(int(keep_largest_component)), L.ptr(binaries), L.ptr(thre), L.ptr(tmp), L.stream()),
            "ia_occgrid_binarize")
    return binaries, thre


def _meshgrid3d(res, device):
    r = [int(v) for v in res]
    return torch.stack(torch.meshgrid([torch.arange(r[0]), torch.arange(r[1]), torch.arange(r[2])], indexing="ij"), -1).to(device)


class TemporalOccGridEstimator(torch.nn.Module):
    """`levels` = number of frames; each level a res^3 grid (temporal_occ_grid.py:20-81)."""
    DIM = 3

    def __init__(self, roi_aabb, resolution=64, levels: int = 1):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(-1, 6)
        if roi_aabb.shape[0] == 1 and levels > 1:
            roi_aabb = roi_aabb.expand(levels, 6).clone()
        self.cells_per_lvl = int(resolution.prod())
        self.levels = levels
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", roi_aabb)
        self.register_buffer("occs", torch.zeros(levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))
        self.register_buffer("grid_coords", _meshgrid3d(resolution, "cpu").reshape(self.cells_per_lvl, 3), persistent=False)
        self._bits = {}

    def _grid_bits(self, lvl: int):
        """bit-packed copy of level `lvl` for the traversal kernel, cached per (buffer identity, in-place version):
        load_state_dict() copies into `binaries` in place (version bump), .to(device) / _apply replace the tensor."""
        b = self.binaries
        key = (b.data_ptr(), b._version, str(b.device))
        hit = self._bits.get(lvl)
        if hit is None or hit[0] != key:
            hit = (key, nerfacc.pack_occupancy_bits(b[lvl]))
            self._bits[lvl] = hit
        return hit[1]

    def _apply(self, fn, *a, **kw):
        self._bits = {}
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._bits = {}
        return super()._load_from_state_dict(*a, **kw)

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn=None, alpha_fn=None, near_plane=0.0, far_plane=1e10, t_min=None, t_max=None,
                 t_idx: float = 0.0, render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0, stratified=False,
                 cone_angle=0.0, jitter: Optional[Tensor] = None):
        """temporal_occ_grid.py:84-223 (the visibility branch is dead on this path: sigma_fn/alpha_fn are never passed)."""
        if sigma_fn is not None or alpha_fn is not None:
            raise NotImplementedError("visibility pruning is dead code on the render_step path")
        near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if t_max is not None:
            far_planes = torch.clamp(far_planes, max=t_max)
        if stratified:
            near_planes += (jitter if jitter is not None else torch.rand_like(near_planes)) * render_step_size
        lvl = math.floor(t_idx * self.levels)
        intervals, samples, _ = nerfacc.traverse_grids(rays_o, rays_d, self.binaries[lvl:lvl + 1], self.aabbs[lvl:lvl + 1],
                                                       near_planes=near_planes, far_planes=far_planes,
                                                       step_size=render_step_size, cone_angle=cone_angle,
                                                       grid_bits=self._grid_bits(lvl), termination_planes=False)
        return (intervals, samples.ray_indices) + samples.interval_ends(intervals)

    @torch.no_grad()
    def update_every_n_steps(self, step: int, t_idx: float, occ_eval_fn: Callable, occ_thre: float = 1e-2,
                             ema_decay: float = 0.95, n: int = 16, rand: Optional[Tensor] = None):
        if not self.training:
            raise RuntimeError("You should only call this function only during training. "
                               "Please call _update() directly if you want to update the field during inference.")
        if step % n == 0:
            self._update(step, math.floor(t_idx * self.levels), occ_eval_fn, occ_thre, ema_decay, rand)

    @torch.no_grad()
    def _update(self, step: int, t_idx: int, occ_eval_fn: Callable, occ_thre: float = 0.01, ema_decay: float = 0.95,
                rand: Optional[Tensor] = None):
        """temporal_occ_grid.py:369-411 (all cells of level t_idx)."""
        lvl = t_idx
        dev = self.occs.device
        gc = self.grid_coords.to(dev).float()
        if rand is None:
            rand = torch.rand_like(gc)
        x = (gc + rand) / self.resolution.to(dev)
        x = self.aabbs[lvl, :3] + x * (self.aabbs[lvl, 3:] - self.aabbs[lvl, :3])
        occ = occ_eval_fn(x).reshape(-1).contiguous().float()
        sl = slice(lvl * self.cells_per_lvl, (lvl + 1) * self.cells_per_lvl)
        occs_lvl = self.occs[sl]
        L.check(L.lib().ia_occgrid_ema(L.i64(self.cells_per_lvl), L.ptr(occs_lvl), L.ptr(occ), L.f32(ema_decay), L.stream()),
                "ia_occgrid_ema")
        self.binaries[lvl], _ = binarize(occs_lvl, self.resolution.tolist(), occ_thre, keep_largest_component=True)
        self._bits.pop(lvl, None)


@torch.no_grad()
def compute_test_occupancy_grid(occ_eval_fn: Callable, aabb: Tensor, resolution: int = 64, n_samples: int = 3,
                                occ_thre: float = 0.01, rand: Optional[Tensor] = None):
    """_compute_occupancy_grid + prepare_test_occupancy_grid (intrinsic_avatar.py:307-381):
    n_samples jittered points per voxel -> alpha -> max -> dilate / threshold / largest component.
    returns (occs [res^3], binaries [1,res,res,res])."""
    dev = aabb.device
    gc = _meshgrid3d([resolution] * 3, dev).reshape(-1, 1, 3).float().expand(-1, n_samples, -1)
    if rand is None:
        rand = torch.rand_like(gc)
    x = ((gc + rand) / resolution).reshape(-1, 3)
    x = x * (aabb[3:] - aabb[:3]) + aabb[:3]
    occs = occ_eval_fn(x).reshape(-1, n_samples).max(1)[0]
    binaries, _ = binarize(occs, [resolution] * 3, occ_thre, keep_largest_component=True)
    return occs, binaries[None]
