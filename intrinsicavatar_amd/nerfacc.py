"""Drop-in for the subset of `nerfacc` (0.5.3) that IntrinsicAvatar's render_step imports
(models/intrinsic_avatar.py:20-28, models/occ_grid/temporal_occ_grid.py:8-12,
models/volrend.py:10-14, models/pbr/utils.py:7):

    RayIntervals, RaySamples, traverse_grids, OccGridEstimator,
    render_weight_from_alpha, render_transmittance_from_alpha, accumulate_along_rays

Same names, argument meaning and error behaviour; compute is libia_amd.so (HIP, gfx950).
"""
import math
import os
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from .lib_nerfacc import pack_info as _pack_info_i32


@dataclass
class RayIntervals:
    """nerfacc.data_specs.RayIntervals: packed interval EDGES along rays."""
    vals: Tensor
    packed_info: Optional[Tensor] = None
    ray_indices: Optional[Tensor] = None
    is_left: Optional[Tensor] = None
    is_right: Optional[Tensor] = None

    @property
    def device(self):
        return self.vals.device


class RaySamples:
    """nerfacc.data_specs.RaySamples: packed samples (interval mid-points).  Same constructor arguments and attributes as the dataclass;
    `is_valid` (all true for a grid traversal, read by nothing on the render_step path) is materialised on first read instead of costing
    a fill launch per traversal."""

    def __init__(self, vals: Tensor, packed_info: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                 is_valid: Optional[Tensor] = None, t_starts: Optional[Tensor] = None, t_ends: Optional[Tensor] = None,
                 all_valid: bool = False):
        self.vals, self.packed_info, self.ray_indices = vals, packed_info, ray_indices
        self._is_valid, self._all_valid = is_valid, all_valid
        # extension (None when not produced): the interval ends per sample = intervals.vals[is_left] / intervals.vals[is_right]
        self.t_starts, self.t_ends = t_starts, t_ends

    @property
    def is_valid(self) -> Optional[Tensor]:
        if self._is_valid is None and self._all_valid:
            self._is_valid = torch.ones(self.vals.shape[0], dtype=torch.bool, device=self.vals.device)
        return self._is_valid

    @is_valid.setter
    def is_valid(self, v):
        self._is_valid = v

    @property
    def device(self):
        return self.vals.device

    def interval_ends(self, intervals: "RayIntervals"):
        """(t_starts, t_ends) of the samples: the traversal's own arrays when it wrote them, else the boolean-mask gathers on
        the edge list that the reference does (models/occ_grid/temporal_occ_grid.py sampling, intrinsic_avatar.py:396-428)."""
        if self.t_starts is not None:
            return self.t_starts, self.t_ends
        return intervals.vals[intervals.is_left], intervals.vals[intervals.is_right]


def pack_occupancy_bits(binaries: Tensor) -> Tensor:
    """bool [.., rx, ry, rz] (one level) -> bit-packed uint32 words (int32 storage)."""
    b = binaries.reshape(-1).contiguous()
    if b.dtype != torch.bool and b.dtype != torch.uint8:
        raise ValueError("binaries must be a bool tensor")
    n_cells = b.numel()
    bits = torch.empty((n_cells + 31) // 32, dtype=torch.int32, device=b.device)
    L.check(L.lib().ia_occgrid_pack_bits(L.ptr(b), L.i64(n_cells), L.ptr(bits), L.stream()), "ia_occgrid_pack_bits")
    return bits


@torch.no_grad()
def traverse_grids(
    rays_o: Tensor,  # [n_rays, 3]
    rays_d: Tensor,  # [n_rays, 3]
    binaries: Tensor,  # [m, resx, resy, resz]
    aabbs: Tensor,  # [m, 6]
    near_planes: Optional[Tensor] = None,  # [n_rays]
    far_planes: Optional[Tensor] = None,  # [n_rays]
    step_size: Optional[float] = 1e-3,
    cone_angle: Optional[float] = 0.0,
    grid_bits: Optional[Tensor] = None,  # extension: pre-packed bits of `binaries` (skips re-packing)
    max_extent: Optional[float] = None,  # extension: upper bound of (far - near) over all rays, tightens the output capacity
    method: Optional[str] = None,        # extension: "fused" (single launch, default) | "two_pass"; env IA_TRAVERSE overrides
    incoherent: bool = False,            # extension: hint -- neighbouring rays point anywhere (secondary rays): the fused kernel walks
                                         # each tile's rays in order of their box-crossing span; outputs are identical either way
    termination_planes: bool = True,     # extension: False -> the third result is None and the fused kernel stops a ray where it leaves the
                                         # box of the occupied cells (intervals / samples identical); render_step never reads the planes
) -> Tuple[RayIntervals, RaySamples, Optional[Tensor]]:
    """nerfacc.traverse_grids (call sites temporal_occ_grid.py:166-175, intrinsic_avatar.py:84-93).

    One grid level (m == 1), which is what the reference always passes."""
    if binaries.dim() != 4 or aabbs.dim() != 2 or aabbs.shape[-1] != 6:
        raise ValueError("binaries must be [m,rx,ry,rz] and aabbs [m,6]")
    if binaries.shape[0] != 1 or aabbs.shape[0] != 1:
        raise NotImplementedError("intrinsicavatar_amd.traverse_grids: one grid level per call "
                                  "(the render_step path slices binaries[t_idx:t_idx+1])")
    n_rays = rays_o.shape[0]
    dev = rays_o.device
    rays_o = rays_o.contiguous().float()
    rays_d = rays_d.contiguous().float()
    if near_planes is None:
        near_planes = torch.zeros(n_rays, device=dev)
    if far_planes is None:
        far_planes = torch.full((n_rays,), float("inf"), device=dev)
    near_planes = near_planes.contiguous().float()
    far_planes = far_planes.contiguous().float()
    aabb = aabbs[0].contiguous().float()
    _, rx, ry, rz = binaries.shape
    if grid_bits is None:
        grid_bits = pack_occupancy_bits(binaries[0])
    lib, st = L.lib(), L.stream()
    args = (L.i64(n_rays), L.ptr(rays_o), L.ptr(rays_d), L.ptr(grid_bits), L.i32(rx), L.i32(ry), L.i32(rz), L.ptr(aabb),
            L.ptr(near_planes), L.ptr(far_planes), L.f32(step_size), L.f32(cone_angle))
    # measured on MI355X (tools/microbench.py): the single launch wins for large batches (2M secondary rays: 0.74 vs
    # 1.20 ms), the two-phase protocol for one 540x540 frame of primary rays (0.14 vs 0.16 ms)
    method = method or os.environ.get("IA_TRAVERSE") or ("fused" if n_rays >= FUSED_MIN_RAYS else "two_pass")
    if method == "fused" and n_rays > 0 and step_size > 0 and cone_angle == 0.0:
        out = _traverse_fused(args, n_rays, aabbs[0], step_size, max_extent, dev, incoherent, termination_planes)
        if out is not None:
            return out

    scratch = torch.empty(int(lib.ia_traverse_scratch_bytes(L.i64(n_rays))), dtype=torch.uint8, device=dev)
    pcnt = torch.empty(n_rays, dtype=torch.int64, device=dev)          # n_edges | n_samples << 32
    pstart = torch.empty(n_rays, dtype=torch.int64, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)               # written by the scan
    L.check(lib.ia_traverse_grids_count(*args, L.ptr(scratch), L.ptr(pcnt), st), "ia_traverse_grids_count")
    tmp = L.scan_tmp(n_rays, dev)
    L.check(lib.ia_exclusive_scan_i64(L.ptr(pcnt), L.ptr(pstart), L.ptr(total), L.i64(n_rays), L.ptr(tmp), st), "scan")
    tot = int(total.item())                      # the one host sync of the two-phase protocol
    E, S = tot & 0xFFFFFFFF, tot >> 32

    iv_vals = torch.empty(E, dtype=torch.float32, device=dev)
    iv_flags = L.zeros((2, E), dev, torch.bool)
    iv_ray = torch.empty(E, dtype=torch.int64, device=dev)
    sm_vals = torch.empty(S, dtype=torch.float32, device=dev)
    sm_ray = torch.empty(S, dtype=torch.int64, device=dev)
    term = torch.empty(n_rays, dtype=torch.float32, device=dev)
    pinfo = torch.empty((2, n_rays, 2), dtype=torch.int64, device=dev)
    L.check(lib.ia_traverse_grids_fill(*args, L.ptr(scratch), L.ptr(pcnt), L.ptr(pstart), L.ptr(pinfo[0]), L.ptr(pinfo[1]),
                                       L.ptr(iv_vals), L.ptr(iv_flags[0]), L.ptr(iv_flags[1]), L.ptr(iv_ray),
                                       L.ptr(sm_vals), L.ptr(sm_ray), L.ptr(term), st), "ia_traverse_grids_fill")
    intervals = RayIntervals(vals=iv_vals, packed_info=pinfo[0], ray_indices=iv_ray,
                             is_left=iv_flags[0], is_right=iv_flags[1])
    samples = RaySamples(vals=sm_vals, packed_info=pinfo[1], ray_indices=sm_ray, all_valid=True)
    return intervals, samples, (term if termination_planes else None)


FUSED_MIN_RAYS = 1 << 19
FUSED_CAP_PER_RAY = int(os.environ.get("IA_TRAVERSE_CAP_PER_RAY", "24"))
_AABB_DIAG = {}


def _traverse_fused(args, n_rays, aabb, step_size, max_extent, dev, incoherent=False, termination_planes=True):
    """single-launch traversal into capacity-sized buffers (ia_traverse_grids_fused); None = capacity exceeded."""
    key = (aabb.data_ptr(), aabb._version)
    diag = _AABB_DIAG.get(key)
    if diag is None:                     # one tiny D2H copy per grid, not per call (several host threads may get here at once: no read-back of the dict)
        a = aabb.detach().float().cpu()
        diag = float((a[3:] - a[:3]).norm())
        _AABB_DIAG.clear()
        _AABB_DIAG[key] = diag
    extent = diag if max_extent is None else min(diag, float(max_extent))
    smax = int(math.ceil(extent / step_size)) + 2
    # capacity: the worst case is smax samples on every ray; a batch of rays averages far fewer (secondary rays of the headline step: 9 of 66,
    # primary rays 1.2 of 130), and the capacity-sized arrays stay alive behind the returned views.  FUSED_CAP_PER_RAY samples per ray on
    # average (at least 2 M samples) are provided; a batch that needs more raises the kernel's overflow flag and goes through the
    # two-phase protocol (exact sizes) -- same results, 23 instead of 78 GB behind the views of a 16 Mi-ray chunk.
    cap_s = min(n_rays * smax, max(n_rays * FUSED_CAP_PER_RAY, 1 << 21))
    cap_e = cap_s + 8 * n_rays
    if cap_e >= (1 << 31):
        return None
    lib, st = L.lib(), L.stream()
    scratch = torch.empty(int(lib.ia_traverse_fused_scratch_bytes(L.i64(n_rays))), dtype=torch.uint8, device=dev)
    totals = torch.empty(3, dtype=torch.int64, device=dev)
    iv_vals = torch.empty(cap_e, dtype=torch.float32, device=dev)
    iv_flags = torch.empty((2, cap_e), dtype=torch.bool, device=dev)
    iv_ray = torch.empty(cap_e, dtype=torch.int64, device=dev)
    sm_vals = torch.empty(cap_s, dtype=torch.float32, device=dev)
    sm_ray = torch.empty(cap_s, dtype=torch.int64, device=dev)
    sm_ends = torch.empty((2, cap_s), dtype=torch.float32, device=dev)
    term = torch.empty(n_rays, dtype=torch.float32, device=dev) if termination_planes else None
    pinfo = torch.empty((2, n_rays, 2), dtype=torch.int64, device=dev)
    L.check(lib.ia_traverse_grids_fused(*args, L.ptr(scratch), L.i64(cap_e), L.i64(cap_s), L.ptr(totals), L.ptr(pinfo[0]),
                                        L.ptr(pinfo[1]), L.ptr(iv_vals), L.ptr(iv_flags[0]), L.ptr(iv_flags[1]), L.ptr(iv_ray),
                                        L.ptr(sm_vals), L.ptr(sm_ray), L.ptr(term), L.ptr(sm_ends[0]), L.ptr(sm_ends[1]),
                                        L.i32(max(smax, 2) if incoherent else 0), st),
            "ia_traverse_grids_fused")
    E, S, ovf = (int(v) for v in totals.tolist())          # the one host sync (output sizes are data dependent)
    if ovf:
        return None
    intervals = RayIntervals(vals=iv_vals[:E], packed_info=pinfo[0], ray_indices=iv_ray[:E],
                             is_left=iv_flags[0, :E], is_right=iv_flags[1, :E])
    samples = RaySamples(vals=sm_vals[:S], packed_info=pinfo[1], ray_indices=sm_ray[:S], all_valid=True,
                         t_starts=sm_ends[0, :S], t_ends=sm_ends[1, :S])
    return intervals, samples, term


# ----------------------------------------------------------------------------- compositing
def _packed_info_i32(packed_info, ray_indices, n_rays, n_samples):
    if packed_info is not None:
        return packed_info.to(torch.int32).contiguous()
    if ray_indices is None or n_rays is None:
        raise ValueError("either packed_info or (ray_indices, n_rays) must be given for packed inputs")
    return _pack_info_i32(ray_indices, n_rays)


class _WeightFromAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas, packed_info):
        alphas = alphas.contiguous().float()
        w = torch.empty_like(alphas)
        t = torch.empty_like(alphas)
        L.check(L.lib().ia_render_weight_from_alpha(L.i64(packed_info.shape[0]), L.ptr(packed_info), L.ptr(alphas),
                                                    L.ptr(w), L.ptr(t), L.stream()), "ia_render_weight_from_alpha")
        ctx.save_for_backward(alphas, packed_info, w, t)
        ctx.set_materialize_grads(False)          # an unused output's gradient arrives as None (NULL in the kernel), not as a zero fill
        return w, t

    @staticmethod
    def backward(ctx, gw, gt):
        alphas, packed_info, w, t = ctx.saved_tensors
        gw = gw.contiguous().float() if gw is not None else None
        gt = gt.contiguous().float() if gt is not None else None
        ga = torch.empty_like(alphas)
        L.check(L.lib().ia_render_weight_from_alpha_bwd(
            L.i64(packed_info.shape[0]), L.ptr(packed_info), L.ptr(alphas), L.ptr(w), L.ptr(t), L.ptr(gw), L.ptr(gt),
            L.ptr(ga), L.stream()), "ia_render_weight_from_alpha_bwd")
        return ga, None


def render_weight_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None,
                             ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None
                             ) -> Tuple[Tensor, Tensor]:
    """nerfacc.render_weight_from_alpha: w_i = T_i a_i, T_i = prod_{j<i}(1-a_j) (packed inputs).
    Differentiable w.r.t. alphas.  Returns (weights, trans)."""
    if alphas.dim() != 1:
        raise NotImplementedError("only packed (flattened) inputs are on the render_step path")
    pi = _packed_info_i32(packed_info, ray_indices, n_rays, alphas.shape[0])
    return _WeightFromAlpha.apply(alphas, pi)


def render_transmittance_from_alpha(alphas, packed_info=None, ray_indices=None, n_rays=None):
    return render_weight_from_alpha(alphas, packed_info, ray_indices, n_rays)[1]


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Optional[Tensor] = None,
                               ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None
                               ) -> Tuple[Tensor, Tensor, Tensor]:
    """nerfacc.render_weight_from_density (imported by models/volrend.py:10-14, density-based `rendering` variant):
    alpha_i = 1 - exp(-sigma_i (t_end_i - t_start_i)), then the alpha compositing kernel.  Returns (weights, trans, alphas).
    Not exercised by the SDF path of render_step (Laplace density -> get_alpha); parity unpinned (nerfacc is not in tree)."""
    alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
    w, tr = render_weight_from_alpha(alphas, packed_info, ray_indices, n_rays)
    return w, tr, alphas


def render_visibility_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                                 n_rays: Optional[int] = None, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0) -> Tensor:
    """nerfacc.render_visibility_from_alpha (imported at models/intrinsic_avatar.py:24, occ_grid/temporal_occ_grid.py:10;
    its call site is behind `alpha_fn is not None`, which the model never passes): visible = T >= early_stop_eps
    [and alpha >= alpha_thre]."""
    trans = render_transmittance_from_alpha(alphas.detach(), packed_info, ray_indices, n_rays)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


def render_visibility_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Optional[Tensor] = None,
                                   ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                                   early_stop_eps: float = 1e-4, alpha_thre: float = 0.0) -> Tensor:
    """nerfacc.render_visibility_from_density: the same test on alphas derived from densities."""
    alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
    return render_visibility_from_alpha(alphas, packed_info, ray_indices, n_rays, early_stop_eps, alpha_thre)


class _Accumulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, values, ray_indices, packed_info):
        weights = weights.contiguous().float()
        n_rays = packed_info.shape[0]
        dim = 1 if values is None else values.shape[-1]
        if values is not None:
            values = values.contiguous().float()
        if weights.shape[0] == 0:       # no samples at all (an empty tensor has a NULL data pointer, which the C ABI reads
            out = torch.zeros((n_rays, dim), dtype=torch.float32, device=weights.device)      # as "values omitted")
        else:
            out = torch.empty((n_rays, dim), dtype=torch.float32, device=weights.device)
            L.check(L.lib().ia_accumulate_along_rays(L.i64(n_rays), L.ptr(packed_info), L.i32(dim), L.ptr(weights),
                                                     L.ptr(values), L.ptr(out), L.stream()), "ia_accumulate_along_rays")
        ctx.save_for_backward(weights, values, ray_indices)
        ctx.dim = dim
        return out

    @staticmethod
    def backward(ctx, g):
        weights, values, ray_indices = ctx.saved_tensors
        g = g.contiguous().float()
        need_w, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and values is not None
        gw = torch.empty_like(weights) if need_w else None
        gv = torch.empty_like(values) if need_v else None
        if weights.shape[0] == 0:
            return gw, gv, None, None
        L.check(L.lib().ia_accumulate_along_rays_bwd(
            L.i64(weights.shape[0]), L.i32(ctx.dim), L.ptr(ray_indices), L.ptr(weights), L.ptr(values), L.ptr(g),
            L.ptr(gw), L.ptr(gv), L.stream()), "ia_accumulate_along_rays_bwd")
        return gw, gv, None, None


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                          n_rays: Optional[int] = None) -> Tensor:
    """nerfacc.accumulate_along_rays: out[r] = sum_{i: ray_i = r} w_i v_i  ([n_rays, D]; D = 1 if values is None).
    Samples must be sorted by ray (always true on the render_step path)."""
    if weights.dim() != 1:
        raise NotImplementedError("only packed (flattened) inputs are on the render_step path")
    if ray_indices is None or n_rays is None:
        raise ValueError("ray_indices and n_rays are required for packed inputs")
    if values is not None and (values.dim() != 2 or values.shape[0] != weights.shape[0]):
        raise ValueError("values must be [n_samples, D]")
    ray_indices = ray_indices.contiguous()
    pi = _pack_info_i32(ray_indices, n_rays)
    return _Accumulate.apply(weights, values, ray_indices, pi)


# ----------------------------------------------------------------------------- estimator shell
class OccGridEstimator(torch.nn.Module):
    """Buffer-compatible shell of nerfacc.OccGridEstimator (buffers `resolution`, `aabbs`, `occs`,
    `binaries`; patchable `.sampling`), as constructed at models/intrinsic_avatar.py:374-379."""
    DIM: int = 3

    def __init__(self, roi_aabb, resolution=128, levels: int = 1, **kwargs):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * self.DIM
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(-1)
        assert roi_aabb.numel() == 6 and levels == 1, "one-level grids only on this path"
        self.cells_per_lvl = int(resolution.prod().item())
        self.levels = levels
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", roi_aabb[None])
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn: Optional[Callable] = None, alpha_fn: Optional[Callable] = None,
                 near_plane: float = 0.0, far_plane: float = 1e10, t_min=None, t_max=None,
                 render_step_size: float = 1e-3, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
                 stratified: bool = False, cone_angle: float = 0.0):
        """== sampling_override, models/intrinsic_avatar.py:49-141 (sigma_fn/alpha_fn are never passed
        on the render_step path; the visibility branch is therefore not implemented)."""
        if sigma_fn is not None or alpha_fn is not None:
            raise NotImplementedError("visibility pruning is dead code on the render_step path")
        near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if t_max is not None:
            far_planes = torch.clamp(far_planes, max=t_max)
        if stratified:
            near_planes += torch.rand_like(near_planes) * render_step_size
        intervals, samples, _ = traverse_grids(rays_o, rays_d, self.binaries, self.aabbs, near_planes=near_planes,
                                               far_planes=far_planes, step_size=render_step_size,
                                               cone_angle=cone_angle, termination_planes=False)
        t_starts, t_ends = samples.interval_ends(intervals)
        return intervals, samples.ray_indices, t_starts, t_ends
