"""Drop-in for the reference's customised `lib.nerfacc` package (lib/nerfacc/__init__.py:5-23):

    ray_resampling, ray_resampling_merge, ray_resampling_fine, ray_resampling_sdf_fine,
    pack_info, pack_data, unpack_info, unpack_data

Signatures, dtypes, shapes and zero/-1 initialisation of outputs follow lib/nerfacc/cdf.py,
lib/nerfacc/pack.py and the host launchers in lib/nerfacc/cuda/csrc/{cdf,pack}.cu.
Compute is libia_amd.so (HIP, gfx950).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L


def _i32c(t: Tensor) -> Tensor:
    if t.dim() != 2 or t.shape[-1] != 2:
        raise RuntimeError("packed_info must be a 2D tensor with shape (n_rays, 2)")
    return t.to(torch.int32).contiguous()


def _resample_info(packed_info: Tensor, n: int, add_steps: bool, read_total: bool = True) -> Tuple[Tensor, int]:
    """resampled packed_info + the total (the .item() of cdf.cu:183).  read_total False: no read-back -- the caller sizes its outputs for
    the worst case n x n_rays and only ever walks them through the returned packed_info; the second result is that capacity."""
    n_rays = packed_info.shape[0]
    dev = packed_info.device
    rpi = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)          # written by the scan
    tmp = L.scan_tmp(n_rays, dev, extra_bytes=8 * n_rays + 64)
    L.check(L.lib().ia_resample_packed_info(L.i64(n_rays), L.ptr(packed_info), L.i32(n), L.i32(int(add_steps)),
                                            L.ptr(rpi), L.ptr(total), L.ptr(tmp), L.stream()),
            "ia_resample_packed_info")
    if not read_total:
        return rpi, n * n_rays
    return rpi, int(total.item())     # the .item() of cdf.cu:183


def _f32v(t: Tensor, n: Optional[int] = None) -> Tensor:
    return t.contiguous().float()


def _resample_tmp(n_rays: int, n_in: int, n: int, dev) -> Tensor:
    """scratch of one K1..K4 call: u-table, CDF tables, per-ray records (csrc/resample.hip)."""
    nbytes = int(L.lib().ia_resample_tmp_bytes(L.i64(n_rays), L.i64(n_in), L.i32(n)))
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


# ----------------------------------------------------------------------------- K1
@torch.no_grad()
def ray_resampling(packed_info: Tensor, t_starts: Tensor, t_ends: Tensor, weights: Tensor, sdfs: Tensor,
                   n_samples: int):
    """lib/nerfacc/cdf.py:13-76 -> cdf.cu:10-215."""
    assert n_samples > 1  # the kernel does not handle n_samples == 1 (cdf.py:49)
    packed_info = _i32c(packed_info)
    if t_starts.dim() != 2 or t_starts.shape[1] != 1 or t_ends.dim() != 2 or t_ends.shape[1] != 1:
        raise RuntimeError("starts/ends must have shape (n_samples_in, 1)")
    if weights.dim() != 1:
        raise RuntimeError("weights must be 1D")
    st, en, w, sd = _f32v(t_starts), _f32v(t_ends), _f32v(weights), _f32v(sdfs)
    n_rays, dev = packed_info.shape[0], packed_info.device
    rpi, T = _resample_info(packed_info, n_samples, False)
    ts = torch.empty((T, 1), dtype=torch.float32, device=dev)
    offs = torch.empty((T, 1), dtype=torch.float32, device=dev)
    idxs = torch.empty((T,), dtype=torch.int64, device=dev)
    # every element of every output is written by the kernels (no -1 / zero fills)
    surface_idx = torch.empty((n_rays,), dtype=torch.int64, device=dev)
    fg = torch.empty((w.shape[0],), dtype=torch.int32, device=dev)
    bg = torch.empty((n_rays,), dtype=torch.int32, device=dev)
    tmp = _resample_tmp(n_rays, w.shape[0], n_samples, dev)
    L.check(L.lib().ia_ray_resampling(L.i64(n_rays), L.i64(w.shape[0]), L.i32(n_samples), L.ptr(packed_info), L.ptr(st), L.ptr(en),
                                      L.ptr(w), L.ptr(sd), L.ptr(rpi), L.i64(T), L.ptr(ts), L.ptr(offs), L.ptr(surface_idx), L.ptr(idxs),
                                      L.ptr(fg), L.ptr(bg), L.ptr(tmp), L.stream()), "ia_ray_resampling")
    return rpi, ts, offs, idxs, fg, bg, surface_idx


@torch.no_grad()
def ray_resampling_capacity(packed_info: Tensor, t_starts: Tensor, t_ends: Tensor, weights: Tensor, sdfs: Tensor, n_samples: int,
                            total: Tensor):
    """ray_resampling without its size read-back (extension; cdf.cu:183): the per-resample outputs are sized for
    n_samples x n_rays slots (every ray hit), the true total T = n_samples x (rays with samples) is written to `total` (int32 [1], on the
    device) and bounds the per-output phase there (ia_ray_resampling_upto); slots behind T stay unwritten.  The caller reads `total` with
    its next size and takes [:T] of ts / offsets / indices (pbr.VolumeInteraction: together with the foreground count).
    -> the tuple of ray_resampling, the three per-resample tensors at capacity."""
    assert n_samples > 1
    packed_info = _i32c(packed_info)
    st, en, w, sd = _f32v(t_starts), _f32v(t_ends), _f32v(weights), _f32v(sdfs)
    n_rays, dev = packed_info.shape[0], packed_info.device
    cap = int(n_samples) * n_rays
    assert cap < (1 << 31), "n_samples x n_rays must stay below 2^31"
    rpi = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    L.check(L.lib().ia_resample_packed_info(L.i64(n_rays), L.ptr(packed_info), L.i32(n_samples), L.i32(0), L.ptr(rpi), L.ptr(total),
                                            L.ptr(L.scan_tmp(n_rays, dev, extra_bytes=8 * n_rays + 64)), L.stream()), "ia_resample_packed_info")
    ts = torch.empty((cap, 1), dtype=torch.float32, device=dev)
    offs = torch.empty((cap, 1), dtype=torch.float32, device=dev)
    idxs = torch.empty((cap,), dtype=torch.int64, device=dev)
    surface_idx = torch.empty((n_rays,), dtype=torch.int64, device=dev)
    fg = torch.empty((w.shape[0],), dtype=torch.int32, device=dev)
    bg = torch.empty((n_rays,), dtype=torch.int32, device=dev)
    tmp = _resample_tmp(n_rays, w.shape[0], n_samples, dev)
    L.check(L.lib().ia_ray_resampling_upto(L.i64(n_rays), L.i64(w.shape[0]), L.i32(n_samples), L.ptr(packed_info), L.ptr(st), L.ptr(en),
                                           L.ptr(w), L.ptr(sd), L.ptr(rpi), L.i64(cap), L.ptr(total), L.ptr(ts), L.ptr(offs),
                                           L.ptr(surface_idx), L.ptr(idxs), L.ptr(fg), L.ptr(bg), L.ptr(tmp), L.stream()), "ia_ray_resampling_upto")
    return rpi, ts, offs, idxs, fg, bg, surface_idx


# ----------------------------------------------------------------------------- K2
@torch.no_grad()
def ray_resampling_merge(packed_info: Tensor, vals: Tensor, is_left: Tensor, is_right: Tensor, weights: Tensor,
                         n_samples: int):
    """lib/nerfacc/cdf.py:79-141 -> cdf.cu:217-401."""
    packed_info = _i32c(packed_info)
    for t in (vals, is_left, is_right, weights):
        if t.dim() != 1:
            raise RuntimeError("vals/is_left/is_right/weights must be 1D")
    vals, w = _f32v(vals), _f32v(weights)
    il, ir = is_left.contiguous(), is_right.contiguous()
    if il.dtype != torch.bool or ir.dtype != torch.bool:
        raise RuntimeError("is_left/is_right must be bool")
    n_rays, dev = packed_info.shape[0], packed_info.device
    rpi, T = _resample_info(packed_info, n_samples, True)
    f = torch.empty((2, T), dtype=torch.float32, device=dev)
    b = torch.empty((4, T), dtype=torch.bool, device=dev)
    tmp = _resample_tmp(n_rays, vals.shape[0], n_samples, dev)
    L.check(L.lib().ia_ray_resampling_merge(L.i64(n_rays), L.i64(vals.shape[0]), L.i32(n_samples), L.ptr(packed_info), L.ptr(vals),
                                            L.ptr(il), L.ptr(ir), L.ptr(w), L.ptr(rpi), L.ptr(f[0]), L.ptr(f[1]), L.ptr(b[0]), L.ptr(b[1]),
                                            L.ptr(b[2]), L.ptr(b[3]), L.ptr(tmp), L.stream()), "ia_ray_resampling_merge")
    # (resampled_packed_info, vals, dists, is_left, is_right, is_resampled, is_fg_sample)
    return rpi, f[0], f[1], b[0], b[1], b[2], b[3]


@torch.no_grad()
def ray_resampling_merge_compact(packed_info: Tensor, vals: Tensor, is_left: Tensor, is_right: Tensor, weights: Tensor, n_samples: int):
    """ray_resampling_merge followed by what its caller does with the result (models/intrinsic_avatar.py:1221-1226):
        keep = is_fg_sample;  vals[keep], is_left[keep], is_right[keep];  ray_indices = unpack_info(rpi)[keep];  pack_info(ray_indices)
    fused (ia_ray_resampling_merge_count / _fill): a ray's reached edges are a prefix of its range, so a scan over RAYS places every
    kept edge directly -- no zero fill, no nonzero, no gathers; one size read-back.
    -> (vals [T], is_left [T], is_right [T], ray_indices int64 [T], packed_info int32 [n_rays, 2])."""
    packed_info = _i32c(packed_info)
    vals, w = _f32v(vals), _f32v(weights)
    il, ir = is_left.contiguous(), is_right.contiguous()
    if il.dtype != torch.bool or ir.dtype != torch.bool:
        raise RuntimeError("is_left/is_right must be bool")
    n_rays, dev = packed_info.shape[0], packed_info.device
    lib, st = L.lib(), L.stream()
    cnt, start = (torch.empty(n_rays, dtype=torch.int32, device=dev) for _ in range(2))
    total = torch.empty(1, dtype=torch.int32, device=dev)          # written by the scan
    tmp = _resample_tmp(n_rays, vals.shape[0], n_samples, dev)
    L.check(lib.ia_ray_resampling_merge_count(L.i64(n_rays), L.i64(vals.shape[0]), L.i32(n_samples), L.ptr(packed_info), L.ptr(vals),
                                              L.ptr(il), L.ptr(ir), L.ptr(w), L.ptr(cnt), L.ptr(start), L.ptr(total), L.ptr(tmp),
                                              L.ptr(L.scan_tmp(n_rays, dev)), st), "ia_ray_resampling_merge_count")
    T = int(total.item())
    ov = torch.empty(T, dtype=torch.float32, device=dev)
    ol, orr = torch.empty(T, dtype=torch.bool, device=dev), torch.empty(T, dtype=torch.bool, device=dev)
    ray = torch.empty(T, dtype=torch.int64, device=dev)
    pinfo = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    L.check(lib.ia_ray_resampling_merge_fill(L.i64(n_rays), L.i64(vals.shape[0]), L.i32(n_samples), L.ptr(packed_info), L.ptr(vals),
                                             L.ptr(il), L.ptr(ir), L.ptr(cnt), L.ptr(start), L.ptr(ov), L.ptr(ol), L.ptr(orr), L.ptr(ray),
                                             L.ptr(pinfo), L.ptr(tmp), st), "ia_ray_resampling_merge_fill")
    return ov, ol, orr, ray, pinfo


# ----------------------------------------------------------------------------- K3 / K4
def _fine(packed_info, t_starts, t_ends, wa, sdfs, n_samples, sdf_mode, exact_size=True):
    packed_info = _i32c(packed_info)
    if t_starts.dim() != 2 or t_starts.shape[1] != 1 or t_ends.dim() != 2 or t_ends.shape[1] != 1:
        raise RuntimeError("starts/ends must have shape (n_samples_in, 1)")
    if wa.dim() != 1:
        raise RuntimeError("weights/alphas must be 1D")
    st, en, wa = _f32v(t_starts), _f32v(t_ends), _f32v(wa)
    n_rays, dev = packed_info.shape[0], packed_info.device
    # exact_size False (extension; few points per ray only: the per-ray kernel never needs the total): the outputs are sized for the worst
    # case n x n_rays, their tail past the last ray's range stays unwritten, and the size read-back of cdf.cu:511 is skipped -- for callers
    # that walk the result through the returned packed_info (compact_foreground does)
    capacity = (not exact_size) and n_samples <= 8
    rpi, T = _resample_info(packed_info, n_samples, False, read_total=not capacity)
    rs = torch.empty((T, 1), dtype=torch.float32, device=dev)
    re = torch.empty((T, 1), dtype=torch.float32, device=dev)
    fg = torch.empty((T,), dtype=torch.bool, device=dev)
    tmp = _resample_tmp(n_rays, wa.shape[0], n_samples, dev) if n_samples > 8 else None      # few points per ray stay in registers
    if sdf_mode:
        sd = _f32v(sdfs)
        L.check(L.lib().ia_ray_resampling_sdf_fine(L.i64(n_rays), L.i64(wa.shape[0]), L.i32(n_samples), L.ptr(packed_info), L.ptr(st),
                                                   L.ptr(en), L.ptr(wa), L.ptr(sd), L.ptr(rpi), L.i64(T), L.ptr(rs), L.ptr(re), L.ptr(fg),
                                                   L.ptr(tmp), L.stream()), "ia_ray_resampling_sdf_fine")
    else:
        L.check(L.lib().ia_ray_resampling_fine(L.i64(n_rays), L.i64(wa.shape[0]), L.i32(n_samples), L.ptr(packed_info), L.ptr(st),
                                               L.ptr(en), L.ptr(wa), L.ptr(rpi), L.i64(T), L.ptr(rs), L.ptr(re), L.ptr(fg), L.ptr(tmp),
                                               L.stream()), "ia_ray_resampling_fine")
    return rpi, rs, re, fg


@torch.no_grad()
def ray_resampling_fine(packed_info, t_starts, t_ends, weights, n_samples):
    """lib/nerfacc/cdf.py:198-244 -> cdf.cu:403-534."""
    return _fine(packed_info, t_starts, t_ends, weights, None, n_samples, False)


@torch.no_grad()
def ray_resampling_sdf_fine(packed_info, t_starts, t_ends, alphas, sdfs, n_samples, exact_size=True):
    """lib/nerfacc/cdf.py:144-195 -> cdf.cu:536-696.  exact_size (extension): see _fine."""
    return _fine(packed_info, t_starts, t_ends, alphas, sdfs, n_samples, True, exact_size=exact_size)


@torch.no_grad()
def compact_foreground(resampled_packed_info: Tensor, starts: Tensor, ends: Tensor, is_fg: Tensor):
    """what the caller of ray_resampling_fine / _sdf_fine does with the result (models/intrinsic_avatar.py:516-528):
        ray_indices = unpack_info(packed_info, T)[is_fg];  t_starts = starts[is_fg];  t_ends = ends[is_fg];  pack_info(ray_indices)
    as a count -> scan over rays -> segmented copy (ia_fg_count / ia_fg_compact): one size read-back instead of three.
    -> (ray_indices int64 [F], t_starts [F], t_ends [F], packed_info int32 [n_rays, 2])."""
    n_rays = resampled_packed_info.shape[0]
    dev = starts.device
    rpi = resampled_packed_info.to(torch.int32).contiguous()
    fg = is_fg.contiguous()
    st_, en_ = starts.reshape(-1).contiguous().float(), ends.reshape(-1).contiguous().float()
    lib, st = L.lib(), L.stream()
    cnt, start = (torch.empty(n_rays, dtype=torch.int32, device=dev) for _ in range(2))
    total = torch.empty(1, dtype=torch.int32, device=dev)          # written by the scan
    L.check(lib.ia_fg_count(L.i64(n_rays), L.ptr(rpi), L.ptr(fg), L.ptr(cnt), L.ptr(start), L.ptr(total), L.ptr(L.scan_tmp(n_rays, dev)), st),
            "ia_fg_count")
    F_ = int(total.item())
    ray_indices = torch.empty(F_, dtype=torch.int64, device=dev)
    ts, te = torch.empty(F_, device=dev), torch.empty(F_, device=dev)
    pinfo = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    L.check(lib.ia_fg_compact(L.i64(n_rays), L.ptr(rpi), L.ptr(fg), L.ptr(st_), L.ptr(en_), L.ptr(cnt), L.ptr(start), L.ptr(ray_indices),
                              L.ptr(ts), L.ptr(te), L.ptr(pinfo), st), "ia_fg_compact")
    return ray_indices, ts, te, pinfo


class IntervalSamples:
    """the samples of an interval (edge) list: t_starts / t_ends [S], ray_indices int64 [S], packed_info int32 [n_rays, 2], and `pos`
    (int32 [n_edges], number of left edges in front of an edge) to carry per-sample values back onto the edges (to_edges)."""

    def __init__(self, is_left, pos, t_starts, t_ends, ray_indices, packed_info):
        self.is_left, self.pos = is_left, pos
        self.t_starts, self.t_ends, self.ray_indices, self.packed_info = t_starts, t_ends, ray_indices, packed_info

    def to_edges(self, sample_vals: Tensor, fill: float) -> Tensor:
        """`x = full(n_edges, fill); x[is_left] = sample_vals` (alpha_fn, models/intrinsic_avatar.py:1022-1025) as a gather."""
        sv = _f32v(sample_vals)
        if sv.shape[0] != self.t_starts.shape[0]:
            raise RuntimeError("to_edges: one value per sample expected")
        out = torch.empty(self.pos.shape[0], dtype=torch.float32, device=sv.device)
        L.check(L.lib().ia_samples_to_edges(L.i64(out.shape[0]), L.ptr(self.is_left), L.ptr(self.pos), L.ptr(sv), L.f32(fill), L.ptr(out),
                                            L.stream()), "ia_samples_to_edges")
        return out


@torch.no_grad()
def interval_samples(packed_info: Tensor, vals: Tensor, is_left: Tensor, ray_indices: Tensor) -> IntervalSamples:
    """what forward_ does with a RayIntervals after every re-sampling (models/intrinsic_avatar.py:1242-1247):
        t_starts = vals[is_left];  t_ends = vals[is_right];  ray_indices = ray_indices[is_left];  pack_info(ray_indices)
    as flag -> scan -> fill over the edges (ia_interval_samples_count / _fill): one size read-back instead of a nonzero, three gathers
    and pack_info's own."""
    packed_info = _i32c(packed_info)
    vals = _f32v(vals)
    il = is_left.contiguous()
    if il.dtype != torch.bool:
        raise RuntimeError("is_left must be bool")
    ray_indices = ray_indices.contiguous()
    if ray_indices.dtype != torch.int64:
        raise RuntimeError("ray_indices must be int64")
    n_rays, n_edges, dev = packed_info.shape[0], vals.shape[0], vals.device
    if il.shape[0] != n_edges or ray_indices.shape[0] != n_edges:
        raise RuntimeError("vals, is_left and ray_indices must have one entry per edge")
    lib, st = L.lib(), L.stream()
    pos = torch.empty(n_edges, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)          # written by the scan
    L.check(lib.ia_interval_samples_count(L.i64(n_edges), L.ptr(il), L.ptr(pos), L.ptr(total), L.ptr(L.scan_tmp(n_edges, dev)), st),
            "ia_interval_samples_count")
    S = int(total.item())
    ts, te = torch.empty(S, dtype=torch.float32, device=dev), torch.empty(S, dtype=torch.float32, device=dev)
    ray = torch.empty(S, dtype=torch.int64, device=dev)
    pinfo = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    L.check(lib.ia_interval_samples_fill(L.i64(n_rays), L.i64(n_edges), L.ptr(packed_info), L.ptr(vals), L.ptr(ray_indices), L.ptr(il),
                                         L.ptr(pos), L.ptr(total), L.ptr(None), L.ptr(ts), L.ptr(te), L.ptr(ray), L.ptr(pinfo), st),
            "ia_interval_samples_fill")
    return IntervalSamples(il, pos, ts, te, ray, pinfo)


@torch.no_grad()
def ray_resampling_merge_compact_samples(packed_info: Tensor, vals: Tensor, is_left: Tensor, is_right: Tensor, weights: Tensor, n_samples: int):
    """ray_resampling_merge_compact + interval_samples of its result with ONE size read-back for both (the reference
    takes `.item()` at cdf.cu:370 and a nonzero for the samples): the kept edges go into CAPACITY-sized buffers -- a ray with edges keeps at
    most all of them plus n_samples new ones, so n_in + n_samples * n_rays slots always suffice --, their number T stays on the device, the
    left-edge scan runs over the capacity with the slots behind T counting nothing (ia_interval_samples_count_upto), and (T, S) come back
    together.  The returned edge tensors are [:T] views of the capacity buffers.
    -> (vals [T], is_left [T], is_right [T], ray_indices int64 [T], packed_info int32 [n_rays, 2], IntervalSamples of that list)."""
    packed_info = _i32c(packed_info)
    vals, w = _f32v(vals), _f32v(weights)
    il, ir = is_left.contiguous(), is_right.contiguous()
    if il.dtype != torch.bool or ir.dtype != torch.bool:
        raise RuntimeError("is_left/is_right must be bool")
    n_rays, n_in, dev = packed_info.shape[0], vals.shape[0], packed_info.device
    cap = n_in + int(n_samples) * n_rays
    if cap == 0 or cap >= (1 << 31):
        ov, ol, orr, ray, pinfo = ray_resampling_merge_compact(packed_info, vals, il, ir, w, n_samples)
        return ov, ol, orr, ray, pinfo, interval_samples(pinfo, ov, ol, ray)
    lib, st = L.lib(), L.stream()
    cnt, start = (torch.empty(n_rays, dtype=torch.int32, device=dev) for _ in range(2))
    totals = torch.empty(2, dtype=torch.int32, device=dev)          # [T, S], both written by scans
    tmp = _resample_tmp(n_rays, n_in, n_samples, dev)
    L.check(lib.ia_ray_resampling_merge_count(L.i64(n_rays), L.i64(n_in), L.i32(n_samples), L.ptr(packed_info), L.ptr(vals),
                                              L.ptr(il), L.ptr(ir), L.ptr(w), L.ptr(cnt), L.ptr(start), L.ptr(totals[0:1]), L.ptr(tmp),
                                              L.ptr(L.scan_tmp(n_rays, dev)), st), "ia_ray_resampling_merge_count")
    ov = torch.empty(cap, dtype=torch.float32, device=dev)
    ol, orr = torch.empty(cap, dtype=torch.bool, device=dev), torch.empty(cap, dtype=torch.bool, device=dev)
    ray = torch.empty(cap, dtype=torch.int64, device=dev)
    pinfo = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    L.check(lib.ia_ray_resampling_merge_fill(L.i64(n_rays), L.i64(n_in), L.i32(n_samples), L.ptr(packed_info), L.ptr(vals),
                                             L.ptr(il), L.ptr(ir), L.ptr(cnt), L.ptr(start), L.ptr(ov), L.ptr(ol), L.ptr(orr), L.ptr(ray),
                                             L.ptr(pinfo), L.ptr(tmp), st), "ia_ray_resampling_merge_fill")
    pos = torch.empty(cap, dtype=torch.int32, device=dev)
    L.check(lib.ia_interval_samples_count_upto(L.i64(cap), L.ptr(ol), L.ptr(totals[0:1]), L.ptr(pos), L.ptr(totals[1:2]),
                                               L.ptr(L.scan_tmp(cap, dev)), st), "ia_interval_samples_count_upto")
    T, S = totals.tolist()                                          # the one read-back of K2 + the samples of its result
    ov, ol, orr, ray, pos = ov[:T], ol[:T], orr[:T], ray[:T], pos[:T]
    ts, te = torch.empty(S, dtype=torch.float32, device=dev), torch.empty(S, dtype=torch.float32, device=dev)
    sray = torch.empty(S, dtype=torch.int64, device=dev)
    spinfo = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    L.check(lib.ia_interval_samples_fill(L.i64(n_rays), L.i64(T), L.ptr(pinfo), L.ptr(ov), L.ptr(ray), L.ptr(ol), L.ptr(pos), L.ptr(totals[1:2]),
                                         L.ptr(None), L.ptr(ts), L.ptr(te), L.ptr(sray), L.ptr(spinfo), st), "ia_interval_samples_fill")
    return ov, ol, orr, ray, pinfo, IntervalSamples(ol, pos, ts, te, sray, spinfo)


# ----------------------------------------------------------------------------- pack / unpack
def pack_data(data: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """lib/nerfacc/pack.py:12-43 (host-side torch ops in the reference too)."""
    assert data.dim() == 3, "data must be with shape of (n_rays, n_samples, D)."
    assert mask.shape == data.shape[:2], "mask must be with shape of (n_rays, n_samples)."
    assert mask.dtype == torch.bool, "mask must be a boolean tensor."
    packed_data = data[mask]
    num_steps = mask.sum(dim=-1, dtype=torch.int32)
    cum_steps = num_steps.cumsum(dim=0, dtype=torch.int32)
    packed_info = torch.stack([cum_steps - num_steps, num_steps], dim=-1)
    return packed_data, packed_info


@torch.no_grad()
def pack_info(ray_indices: Tensor, n_rays: int = None) -> Tensor:
    """lib/nerfacc/pack.py:46-77: ray_indices (int64, [N]) -> packed_info (int32, [n_rays, 2])."""
    assert ray_indices.dim() == 1, "ray_indices must be a 1D tensor with shape (n_samples)."
    if not ray_indices.is_cuda:
        raise NotImplementedError("Only support cuda inputs.")
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1
    ray_indices = ray_indices.contiguous().to(torch.int64)
    dev = ray_indices.device
    out = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
    tmp = L.scan_tmp(n_rays, dev, extra_bytes=8 * n_rays + 64)
    L.check(L.lib().ia_pack_info(L.i64(ray_indices.shape[0]), L.ptr(ray_indices), L.i64(n_rays), L.ptr(out),
                                 L.ptr(tmp), L.stream()), "ia_pack_info")
    return out


@torch.no_grad()
def unpack_info(packed_info: Tensor, n_samples: int) -> Tensor:
    """lib/nerfacc/pack.py:80-121 -> pack.cu:7-28,84-102."""
    assert packed_info.dim() == 2 and packed_info.shape[-1] == 2, \
        "packed_info must be a 2D tensor with shape (n_rays, 2)."
    if not packed_info.is_cuda:
        raise NotImplementedError("Only support cuda inputs.")
    packed_info = _i32c(packed_info)
    out = torch.empty((n_samples,), dtype=torch.int64, device=packed_info.device)
    L.check(L.lib().ia_unpack_info(L.i64(packed_info.shape[0]), L.ptr(packed_info), L.ptr(out), L.stream()),
            "ia_unpack_info")
    return out


def _unpack_info_to_mask(packed_info: Tensor, n_samples: int) -> Tensor:
    masks = torch.zeros((packed_info.shape[0], n_samples), dtype=torch.bool, device=packed_info.device)
    L.check(L.lib().ia_unpack_info_to_mask(L.i64(packed_info.shape[0]), L.ptr(packed_info), L.i32(n_samples),
                                           L.ptr(masks), L.stream()), "ia_unpack_info_to_mask")
    return masks


class _UnpackData(torch.autograd.Function):
    """lib/nerfacc/pack.py:170-190."""

    @staticmethod
    def forward(ctx, packed_info: Tensor, data: Tensor, n_samples: int):
        packed_info = _i32c(packed_info)
        orig_dtype = data.dtype
        data = data.contiguous()
        if ctx.needs_input_grad[1]:
            ctx.save_for_backward(packed_info)
            ctx.n_samples = n_samples
        # the kernel moves 4-byte words; int64 payloads (the spp shuffle indices,
        # intrinsic_avatar.py:1368-1377) go through as two words per element
        if data.dtype in (torch.float32, torch.int32):
            words, dim = data.view(torch.float32) if data.dtype == torch.int32 else data, data.shape[1]
        elif data.dtype in (torch.int64, torch.float64):
            words, dim = data.view(torch.float32), data.shape[1] * 2
        else:
            raise NotImplementedError(f"unpack_data: dtype {data.dtype}")
        out = torch.zeros((packed_info.shape[0], n_samples, dim), dtype=torch.float32, device=data.device)
        L.check(L.lib().ia_unpack_data(L.i64(packed_info.shape[0]), L.ptr(packed_info), L.i32(dim), L.ptr(words),
                                       L.i32(n_samples), L.ptr(out), L.stream()), "ia_unpack_data")
        return out.view(orig_dtype)

    @staticmethod
    def backward(ctx, grad: Tensor):
        packed_info = ctx.saved_tensors[0]
        mask = _unpack_info_to_mask(packed_info, ctx.n_samples)
        return None, grad[mask].contiguous(), None


def unpack_data(packed_info: Tensor, data: Tensor, n_samples: Optional[int] = None) -> Tensor:
    """lib/nerfacc/pack.py:124-167."""
    assert packed_info.dim() == 2 and packed_info.shape[-1] == 2, \
        "packed_info must be a 2D tensor with shape (n_rays, 2)."
    assert data.dim() == 2, "data must be a 2D tensor with shape (n_samples, D)."
    if n_samples is None:
        n_samples = packed_info[:, 1].max().item()
    return _UnpackData.apply(packed_info, data, n_samples)
