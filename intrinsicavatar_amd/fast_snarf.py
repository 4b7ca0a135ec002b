"""Drop-in for the three fast-SNARF JIT extension modules the reference loads at
models/deformers/fast_snarf/deformer_torch.py:9-18:

    fuse_kernel.fuse_broyden(...)   -> fuse_cuda_kernel_fast.cu:250-452
    filter_cuda.filter(...)         -> filter.cu:10-77
    precompute_cuda.precompute(...) -> precompute.cu:24-103

Same positional signatures (in-place outputs, None return for fuse_broyden / precompute).
`ChannelLastVoxelJ` is the MI355X-native extension: the channel-last copy of voxel_J that
`precompute` can emit for free and `fuse_broyden` gathers 4x more efficiently.
"""
import torch
from torch import Tensor

from . import _lib as L


class ChannelLastVoxelJ:
    """[B,D,H,W,12] fp32 view of the Jacobian grid (48 B per voxel)."""

    def __init__(self, data: Tensor):
        assert data.dim() == 5 and data.shape[-1] == 12
        self.data = data.contiguous()


def precompute(voxel_w: Tensor, tfs: Tensor, voxel_d: Tensor, voxel_J: Tensor, offset: Tensor, scale: Tensor,
               voxel_J_cl: Tensor = None) -> None:
    """precompute_cuda.precompute (deformer_torch.py:86-92). voxel_d / voxel_J written in place.
    Optional extension: voxel_J_cl [B,D,H,W,12] also written (channel-last copy)."""
    B = tfs.shape[0]
    _, C24, D, H, W = voxel_w.shape
    assert C24 == 24 and tfs.shape[1:] == (24, 4, 4)
    voxel_w, tfs = voxel_w.contiguous().float(), tfs.contiguous().float()
    off, sc = offset.reshape(3).contiguous().float(), scale.reshape(3).contiguous().float()
    for t in (voxel_d, voxel_J, voxel_J_cl):
        if t is not None and (not t.is_contiguous() or t.dtype != torch.float32):
            raise RuntimeError("output grids must be contiguous float32")
    L.check(L.lib().ia_precompute(L.i32(B), L.i32(D), L.i32(H), L.i32(W), L.ptr(voxel_w), L.ptr(tfs), L.ptr(off),
                                  L.ptr(sc), L.ptr(voxel_d), L.ptr(voxel_J), L.ptr(voxel_J_cl), L.stream()),
            "ia_precompute")


CELL_TAU = 2.5        # a voxel cell is tight when |(dg/dx)^-1|_F <= CELL_TAU all over it (a rotation has sqrt(3) = 1.73)


def cell_tightness(voxel_J: ChannelLastVoxelJ, offset: Tensor, scale: Tensor, tau: float = CELL_TAU) -> Tensor:
    """uint8 [D,H,W] veto table of the early-filter search (ia_cell_tightness; once per pose, after precompute): bit 0 = the TRUE Jacobian
    of the skinning map -- weight-gradient term included -- keeps its sign and |J^-1|_F <= tau at 27 sample points of the voxel cell
    (bit 2: the sign; bit 1: the 26 neighbouring cells are tight with the same sign, and the retirement box of a root there needs no cut
    to its cell).  A root in a cell without bit 0 retires no search: next to a fold of the map two roots sit 1e-4 ... 1e-3 apart and Broyden's own J_inv
    estimate cannot tell (csrc/snarf.hip cell_tightness_kernel).  No counterpart in the reference."""
    assert isinstance(voxel_J, ChannelLastVoxelJ) and voxel_J.data.shape[0] == 1
    _, D, H, W, _ = voxel_J.data.shape
    out = torch.empty((D, H, W), dtype=torch.uint8, device=voxel_J.data.device)
    L.check(L.lib().ia_cell_tightness(L.i32(D), L.i32(H), L.i32(W), L.ptr(voxel_J.data), L.ptr(offset.reshape(3).contiguous().float()),
                                      L.ptr(scale.reshape(3).contiguous().float()), L.f32(tau), L.ptr(out), L.stream()), "ia_cell_tightness")
    return out


def _cell_tight_arg(cell_tight, voxel_J):
    if cell_tight is None:
        return None
    if cell_tight.dtype != torch.uint8 or tuple(cell_tight.shape) != tuple(voxel_J.data.shape[1:4]) or not cell_tight.is_contiguous():
        raise RuntimeError("cell_tight must be a contiguous uint8 [D,H,W] tensor (fast_snarf.cell_tightness)")
    return cell_tight


def fuse_broyden(x: Tensor, xd_tgt: Tensor, voxel: Tensor, voxel_J, tfs: Tensor, bone_ids: Tensor,
                 align_corners: bool, J_inv: Tensor, is_valid: Tensor, offset: Tensor, scale: Tensor,
                 cvg_threshold: float, dvg_threshold: float, fwd_J: Tensor = None) -> None:
    """fuse_kernel.fuse_broyden (deformer_torch.py:109-121). x, J_inv, is_valid are caller-zeroed
    in/out tensors. `voxel` (voxel_d) and `align_corners` are accepted and ignored, exactly like
    the reference kernel. voxel_J: Tensor [B,12,D,H,W] or ChannelLastVoxelJ."""
    B, N, _ = xd_tgt.shape
    I = bone_ids.shape[0]
    if isinstance(voxel_J, ChannelLastVoxelJ):
        vj, layout = voxel_J.data, 1
        _, D, H, W, _ = vj.shape
    else:
        vj, layout = voxel_J.contiguous(), 0
        _, _, D, H, W = vj.shape
    for t in (x, J_inv, is_valid):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("outputs must be contiguous")
    if x.shape != (B, N, I, 3) or (J_inv is not None and J_inv.shape != (B, N, I, 3, 3)) or is_valid.shape != (B, N, I):
        raise RuntimeError("output shapes must be x[B,N,I,3], J_inv[B,N,I,3,3], is_valid[B,N,I]")
    xd = xd_tgt.contiguous().float()
    tfs = tfs.contiguous().float()
    bones = bone_ids.contiguous().to(torch.int32)
    off, sc = offset.reshape(3).contiguous().float(), scale.reshape(3).contiguous().float()
    L.check(L.lib().ia_fuse_broyden(L.i32(B), L.i64(N), L.i32(I), L.ptr(xd), L.ptr(vj), L.i32(layout), L.i32(D),
                                    L.i32(H), L.i32(W), L.ptr(tfs), L.ptr(bones), L.ptr(off), L.ptr(sc),
                                    L.f32(cvg_threshold), L.f32(dvg_threshold), L.ptr(x), L.ptr(J_inv),
                                    L.ptr(is_valid), L.ptr(fwd_J), L.stream()), "ia_fuse_broyden")


def fuse_broyden_spec(x: Tensor, xd_tgt: Tensor, voxel_J: ChannelLastVoxelJ, tfs: Tensor, bone_ids: Tensor, J_inv: Tensor,
                      is_valid: Tensor, offset: Tensor, scale: Tensor, cvg_threshold: float, dvg_threshold: float, eps: float,
                      fwd_J: Tensor = None, counters: Tensor = None, cell_tight: Tensor = None) -> None:
    """fuse_broyden with the K9-consistent early filter (ia_fuse_broyden_spec; B = 1, channel-last grid): a search that comes within
    `eps` of a TIGHT root found by a LATER init of its point, inside that root's voxel cell, is retired -- K9 (filter.cu:10-54) would
    drop it wherever exactly it ends; points whose completed roots leave K9's decision open are searched again with the filter off
    (csrc/snarf.hip).  Everything that is not retired is bit-identical to fuse_broyden, and filter() of the result equals filter()
    of fuse_broyden's on all but ~1e-7 of the points -- on all points measured (145 M on eight poses) with cell_tight = cell_tightness(...),
    the veto table that keeps roots next to a fold of the skinning map from retiring anything.  No counterpart in the reference; used by
    SNARFDeformer.search.
    counters: optional int64 [5] (accumulated): fetches, retired items, completed valid items, points redone, corner loads."""
    B, N, _ = xd_tgt.shape
    I = bone_ids.shape[0]
    assert B == 1 and isinstance(voxel_J, ChannelLastVoxelJ) and voxel_J.data.shape[0] == 1
    _, D, H, W, _ = voxel_J.data.shape
    if x.shape != (B, N, I, 3) or (J_inv is not None and J_inv.shape != (B, N, I, 3, 3)) or is_valid.shape != (B, N, I):
        raise RuntimeError("output shapes must be x[B,N,I,3], J_inv[B,N,I,3,3], is_valid[B,N,I]")
    for t in (x, J_inv, is_valid, fwd_J):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("outputs must be contiguous")
    L.check(L.lib().ia_fuse_broyden_spec(
        L.i64(N), L.i32(I), L.ptr(xd_tgt.contiguous().float()), L.ptr(voxel_J.data), L.i32(D), L.i32(H), L.i32(W),
        L.ptr(tfs.contiguous().float()), L.ptr(bone_ids.contiguous().to(torch.int32)), L.ptr(offset.reshape(3).contiguous().float()),
        L.ptr(scale.reshape(3).contiguous().float()), L.f32(cvg_threshold), L.f32(dvg_threshold), L.f32(eps), L.ptr(x), L.ptr(J_inv),
        L.ptr(is_valid), L.ptr(fwd_J), L.ptr(counters), L.ptr(_cell_tight_arg(cell_tight, voxel_J)), L.stream()), "ia_fuse_broyden_spec")


def fuse_broyden_spec_rows(x_rows: Tensor, xd_tgt: Tensor, voxel_J: ChannelLastVoxelJ, tfs: Tensor, bone_ids: Tensor, J_inv: Tensor,
                           cnt: Tensor, meta: Tensor, start: Tensor, ovf_head: Tensor, ovf_scratch: Tensor, total_and_overflow: Tensor,
                           offset: Tensor, scale: Tensor, cvg_threshold: float, dvg_threshold: float, eps: float, fwd_J: Tensor = None,
                           counters: Tensor = None, order: Tensor = None, n_points: int = None, cell_tight: Tensor = None) -> None:
    """fuse_broyden_spec with the candidate bookkeeping in the kernel (ia_fuse_broyden_spec_rows): no x [N,I,3], no is_valid, no
    filter pass -- x_rows [N,3,3] receives each point's surviving candidates (highest init first), cnt [N] int32 their number,
    meta [N] int32 their inits (one byte each; bit 31: the point has overflow records), start [N] the exclusive scan of cnt,
    ovf_head [N] int32 + ovf_scratch (uint8 [ia_spec_rows_overflow_bytes()]): the records of the points the kernel redid with the
    filter off and their 4th.. survivors, total_and_overflow [2] int32 = (Q, number of points redone; above
    ia_spec_rows_overflow_capacity() results were lost).  J_inv / fwd_J at [point, init] as fuse_broyden.
    order (int32 [N], optional): point p of the launch is xd_tgt[0, order[p]] -- the caller's points are searched in another
    order (spatially sorted) than they are stored, without a gathered copy."""
    B, N, _ = xd_tgt.shape
    if order is not None:
        N = order.shape[0]
    I = bone_ids.shape[0]
    assert B == 1 and isinstance(voxel_J, ChannelLastVoxelJ) and voxel_J.data.shape[0] == 1
    _, D, H, W, _ = voxel_J.data.shape
    assert x_rows.shape == (N, 3, 3) and all(t.shape == (N,) and t.dtype == torch.int32 for t in (cnt, meta, start, ovf_head))
    assert total_and_overflow.shape == (2,) and total_and_overflow.dtype == torch.int32
    # the work area grows with N (flagged list of max(65536, N / 64) entries): a caller's buffer sized for another N would be overrun
    need = int(L.lib().ia_spec_rows_overflow_bytes(L.i64(N)))
    if ovf_scratch.dtype != torch.uint8 or not ovf_scratch.is_contiguous() or ovf_scratch.numel() < need:
        raise RuntimeError(f"ovf_scratch must be a contiguous uint8 tensor of at least ia_spec_rows_overflow_bytes(N={N}) = {need} bytes "
                           f"(got {ovf_scratch.dtype}, {ovf_scratch.numel()})")
    for t in (x_rows, J_inv, fwd_J, cnt, meta, start):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("outputs must be contiguous")
    L.check(L.lib().ia_fuse_broyden_spec_rows(
        L.i64(N), L.i32(I), L.ptr(xd_tgt.contiguous().float()), L.ptr(voxel_J.data), L.i32(D), L.i32(H), L.i32(W),
        L.ptr(tfs.contiguous().float()), L.ptr(bone_ids.contiguous().to(torch.int32)), L.ptr(offset.reshape(3).contiguous().float()),
        L.ptr(scale.reshape(3).contiguous().float()), L.f32(cvg_threshold), L.f32(dvg_threshold), L.f32(eps), L.ptr(x_rows), L.ptr(J_inv),
        L.ptr(fwd_J), L.ptr(cnt), L.ptr(meta), L.ptr(start), L.ptr(ovf_head), L.ptr(ovf_scratch), L.ptr(total_and_overflow),
        L.ptr(L.scan_tmp(N, xd_tgt.device)), L.ptr(counters), L.ptr(order), L.ptr(_cell_tight_arg(cell_tight, voxel_J)), L.stream()),
        "ia_fuse_broyden_spec_rows")


def filter(x: Tensor, mask: Tensor) -> Tensor:
    """filter_cuda.filter (deformer_torch.py:122, filter.cpp:12-18): drop candidate i if a later valid
    candidate j lies within 1e-4 (keeps the last of a cluster). B must be 1 (filter.cu:21-22)."""
    B, N, I = mask.shape
    if B != 1:
        raise NotImplementedError("filter: B == 1 only (the reference's index math is only valid for B == 1)")
    x = x.contiguous().float()
    mask = mask.contiguous()
    out = torch.empty_like(mask)
    L.check(L.lib().ia_filter(L.i64(N), L.i32(I), L.ptr(x), L.ptr(mask), L.ptr(out), L.stream()), "ia_filter")
    return out
