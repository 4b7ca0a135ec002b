"""Host-side mirror of the reference's SNARF deformer on the MI355X kernels:

  SNARFDeformer (models/deformers/snarf_deformer.py:38-264) + ForwardDeformer
  (models/deformers/fast_snarf/deformer_torch.py:21-197), eval / no-pose-gradient path.

One `deform()` call = Broyden root search for 13 bone initialisations (K8) -> duplicate filter fused
with candidate counting (K9) -> packed candidate list -> SDF network on the candidates -> first-min
select + gather + normal push-forward (ia_deform_select).  The only host sync is the read-back of
the candidate count (the reference syncs twice per call with cudaDeviceSynchronize and once per
boolean-mask index).
"""
from typing import Optional

import ctypes as C
import os
import threading

import torch
from torch import Tensor

from . import _lib as L
from . import fast_snarf


class SNARFDeformer:
    INIT_BONES = [0, 1, 2, 4, 5, 10, 11, 12, 15, 16, 17, 18, 19]    # deformer_torch.py:27
    # K9-consistent early filter of the search (fast_snarf.fuse_broyden_spec[_rows], csrc/snarf.hip): a search is retired
    # once it comes within SPEC_EPS metres of a TIGHT root that a later init of the same point has found, inside that root's voxel
    # cell (K9 would drop it wherever exactly it ends); a point whose completed roots leave K9's decision open is searched again with
    # the filter off; a root in a voxel cell where the TRUE Jacobian of the skinning map is not tight everywhere (cell_tight: per pose,
    # fast_snarf.cell_tightness) retires nothing -- next to a fold of the map Broyden's estimate is blind.  The candidate set equals
    # search-to-the-end + K9 on every point measured (0 of 145 M march points on the eight reference poses,
    # profiles/r04_spec_search_probe_poses.jsonl; without the cell table 3e-7 of them differed), candidates bit-identical.  IA_SPEC_CELL_TAU=0
    # switches the table off (A / B).  On for EVERY batch size: what a point gets must not
    # depend on how many other points share its launch (ray-batch sharding invariance, multi-GPU = single GPU).
    # IA_BROYDEN_SPEC_EPS=0 (or deformer.spec_eps = 0) = the reference's search-to-the-end everywhere; the kernel-level parity tests
    # (K8 / K9 golden vectors) call those entry points directly.
    SPEC_EPS = float(os.environ.get("IA_BROYDEN_SPEC_EPS", "1e-3"))
    SPEC_MIN_POINTS = int(os.environ.get("IA_BROYDEN_SPEC_MIN_POINTS", "1"))

    def __init__(self, lbs_voxel_final: Tensor, offset_kernel: Tensor, scale_kernel: Tensor, bbox: Tensor):
        self.lbs_voxel_final = lbs_voxel_final.contiguous().float()          # [1,24,D,H,W]
        self.offset_kernel = offset_kernel.reshape(3).contiguous().float()
        self.scale_kernel = scale_kernel.reshape(3).contiguous().float()
        self.bbox = bbox                                                     # [2,3] canonical bbox (deformer_torch.py:157)
        self.device = self.lbs_voxel_final.device
        self.init_bones = torch.tensor(self.INIT_BONES, dtype=torch.int32, device=self.device)
        self.spec_eps = self.SPEC_EPS
        self.spec_counters = None            # optional int64 [5] device tensor: accumulated by the early-filter search (bench.py)
        self._tls = threading.local()        # per-call diagnostics of _candidates, per host thread (the secondary march runs it on several)
        self.spec_canary = self.SPEC_CANARY  # every k-th point of a batch is searched again to the end and compared (0 = off)
        self._canary_parts = []              # one int64 [3] device accumulator per stream (canary_totals() adds them up)
        self._canary_by_stream = {}
        self._canary_lock = threading.Lock()
        self.tfs = None
        self.voxel_J_cl = None
        self.cell_tight = None
        self._voxels_finite = None
        self.voxel_d = None
        self.w2s = None

    @property
    def last_overflow_records(self) -> int:
        """points the early-filter kernel searched again with the filter off, in THIS thread's last _candidates call."""
        return getattr(self._tls, "n_over", 0)

    @property
    def _ovf_cap(self) -> int:
        """capacity of the list of such points in this thread's last _candidates call (ia_spec_rows_overflow_capacity(P))."""
        return getattr(self._tls, "ovf_cap", 0)

    # -- per frame -------------------------------------------------------------------
    def prepare(self, tfs: Tensor, w2s: Tensor):
        """prepare_deformer (snarf_deformer.py:81-126) from the bone transforms on: tfs [B,24,4,4] and w2s [4,4] come from
        smpl.SMPLKinematics + smpl.deformer_transforms (or any other rig).  tfs may carry an autograd graph (pose
        optimisation): the kernels read its values, train.shade_differentiable adds the implicit pose terms."""
        _, _, D, H, W = self.lbs_voxel_final.shape
        self.tfs = tfs.contiguous().float()
        self.w2s = w2s.contiguous().float()
        B = self.tfs.shape[0]
        self.voxel_d = torch.empty((B, 3, D, H, W), device=self.device)
        self.voxel_J_cl = torch.empty((B, D, H, W, 12), device=self.device)
        fast_snarf.precompute(self.lbs_voxel_final, self.tfs, self.voxel_d, None, self.offset_kernel, self.scale_kernel,
                              voxel_J_cl=self.voxel_J_cl)
        # the search's corner-pair loads read an out-of-range corner's in-range neighbour with weight 0 (bit-identical to skipping it
        # only for finite voxels): one check per frame instead of a select per load.  The flag stays on the device and is read next
        # to the first size read-back of a search on this pose (_check_voxels) -- no host sync of its own
        self._voxels_finite = torch.isfinite(self.voxel_J_cl).all()
        # veto table of the early filter: cells where the true Jacobian of the skinning map is tight (one small kernel per pose)
        tau = float(os.environ.get("IA_SPEC_CELL_TAU", str(fast_snarf.CELL_TAU)))
        self.cell_tight = (fast_snarf.cell_tightness(fast_snarf.ChannelLastVoxelJ(self.voxel_J_cl), self.offset_kernel, self.scale_kernel, tau)
                           if (tau > 0 and B == 1) else None)

    def _check_voxels(self):
        f = self._voxels_finite
        if f is not None:
            self._voxels_finite = None
            if not bool(f):
                raise RuntimeError("SNARFDeformer.prepare: the skinning grid (voxel_J) holds non-finite values")

    def transform_rays_w2s(self, rays: Tensor) -> Tensor:
        """snarf_deformer.py:128-147: (o R^T + t, d R^T, |o'| - 1, |o'| + 1).  One launch (ia_transform_rays_w2s, the fma-chain form): the
        rotated origins / directions are bit-identical to the [n,3] x [3,3] library products of the torch expression below AND to numpy's
        (the CPU oracle's), the near / far columns to numpy's norm (torch's GPU norm differs from both by an ulp on 3.6 % of the rays) --
        tools/ray_transform_probe.py: 0 of 9.55 M elements differ."""
        w2s = self.w2s
        if not rays.is_cuda:
            raise L.IaError("SNARFDeformer.transform_rays_w2s needs GPU rays (no CPU fallback)")
        if rays.dtype == torch.float32 and rays.dim() == 2 and rays.shape[1] >= 6 and rays.stride(1) == 1 and not rays.requires_grad \
                and not w2s.requires_grad and w2s.shape == (4, 4):
            n = rays.shape[0]
            out = torch.empty((n, 8), device=rays.device)
            L.check(L.lib().ia_transform_rays_w2s(L.i64(n), C.c_void_p(rays.data_ptr()), L.i32(rays.stride(0)), L.ptr(w2s.detach().contiguous()), L.i32(0),
                                                  L.ptr(out), L.stream()), "ia_transform_rays_w2s")
            return out
        rays_o = rays[:, :3] @ w2s[:3, :3].T + w2s[None, :3, 3]
        rays_d = rays[:, 3:6] @ w2s[:3, :3].T
        d = torch.linalg.norm(rays_o, dim=-1, keepdim=True)
        return torch.cat([rays_o, rays_d, d - 1, d + 1], dim=-1)

    # -- per query batch -------------------------------------------------------------
    @torch.no_grad()
    def search(self, pts: Tensor, want_fwd: bool = False, want_jinv: bool = False):
        """K8 for all 13 inits. returns x [P,13,3], valid [P,13] (pre-filter), fwd_J [P,13,3,3] or None
        (, J_inv [P,13,3,3] -- Broyden's inverse-Jacobian estimate at the returned root -- when want_jinv)."""
        P = pts.shape[0]
        I = self.init_bones.shape[0]
        # x / fwd_J are only ever read under the valid mask and the kernel writes is_valid for every item, so none of the
        # three needs the zero-fill the reference API asks of its callers (2.8 GB of fills per 4.4 M-point launch)
        x = torch.empty((1, P, I, 3), device=self.device)
        # use_j_inv: false (configs/deformer/snarf_deformer.yaml:11): inference never reads J_inv and the kernel skips the
        # store; training with pose gradients needs it for the implicit-differentiation correction (deformer_torch.py:57-76)
        Jinv = torch.empty((1, P, I, 3, 3), device=self.device) if want_jinv else None
        valid = torch.empty((1, P, I), dtype=torch.bool, device=self.device)
        fwd = torch.empty((1, P, I, 3, 3), device=self.device) if want_fwd else None
        if self.spec_eps > 0.0 and P >= self.SPEC_MIN_POINTS and self.tfs.shape[0] == 1:
            fast_snarf.fuse_broyden_spec(x, pts.reshape(1, P, 3), fast_snarf.ChannelLastVoxelJ(self.voxel_J_cl), self.tfs,
                                         self.init_bones, Jinv, valid, self.offset_kernel, self.scale_kernel, 1e-5, 1e-1,
                                         self.spec_eps, fwd_J=fwd, counters=self.spec_counters, cell_tight=self.cell_tight)
        else:
            fast_snarf.fuse_broyden(x, pts.reshape(1, P, 3), None, fast_snarf.ChannelLastVoxelJ(self.voxel_J_cl), self.tfs,
                                    self.init_bones, True, Jinv, valid, self.offset_kernel, self.scale_kernel, 1e-5, 1e-1,
                                    fwd_J=fwd)
        if want_jinv:
            return x[0], valid[0], (fwd[0] if want_fwd else None), Jinv[0]
        return x[0], valid[0], (fwd[0] if want_fwd else None)

    def query_weights(self, xc: Tensor) -> Tensor:
        """skinning weights [P,24] at canonical points: trilinear, align_corners, border-clamped lookup of lbs_voxel_final
        (deformer_torch.py:198-209)."""
        g = ((xc + self.offset_kernel) * self.scale_kernel).reshape(1, -1, 1, 1, 3)
        w = torch.nn.functional.grid_sample(self.lbs_voxel_final, g, align_corners=True, mode="bilinear",
                                            padding_mode="border")
        return w.reshape(w.shape[1], -1).t()

    def implicit_pose_terms(self, xc: Tensor, J_inv: Tensor, valid: Tensor):
        """the two places the bone transforms enter the training graph (ForwardDeformer.forward, version 1,
        deformer_torch.py:57-76, and SNARFDeformer.deform_, snarf_deformer.py:176-184), for the winning roots only:
          * xc + correction, correction = -J_inv (LBS(xc, tfs) - stopgrad(LBS(xc, tfs))): zero in value, and its derivative
            with respect to tfs is the implicit-function derivative of the root;
          * the linearly blended rotation block T[:, :3, :3] that pushes normals to observation space.
        xc, J_inv, valid are constants (no_grad search results); self.tfs carries the graph."""
        with torch.no_grad():
            w = self.query_weights(xc)                                          # [P,24]
        T = (w @ self.tfs[0].reshape(w.shape[1], 16)).reshape(-1, 4, 4)
        R = T[:, :3, :3]
        xd = (R * xc[:, None, :]).sum(-1) + T[:, :3, 3]
        corr = -(J_inv * (xd - xd.detach())[:, None, :]).sum(-1)
        corr = corr * valid[:, None].to(corr.dtype)
        return xc + corr, R

    @torch.no_grad()
    def _pack_candidates(self, x: Tensor, valid: Tensor, with_src: bool):
        """K9 filter + per-point count + exclusive scan + packed candidate list (snarf_deformer.py:187-196's mask indexing).
        x [P,I,3] is consumed.  -> cand_x [Q,3], cand_src [Q] int32 (= p*I+i; None unless with_src), cnt [P], start [P], Q.
        Default: tile-local packing in place + segmented copy (ia_deform_filter_tiles / ia_deform_pack_tiles);
        IA_PACK=lookback: one pass with a chained scan, the packed list a prefix of x's storage (ia_deform_filter_compact)."""
        P, I = valid.shape
        dev = x.device
        lib, st = L.lib(), L.stream()
        cnt = torch.empty(P, dtype=torch.int32, device=dev)
        start = torch.empty(P, dtype=torch.int32, device=dev)
        total = torch.empty(1, dtype=torch.int32, device=dev)
        src = torch.empty(P * I, dtype=torch.int32, device=dev) if with_src else None
        if os.environ.get("IA_PACK", "tiles") == "lookback":
            nbytes = int(lib.ia_deform_filter_compact_tmp_bytes(L.i64(P)))
            tmp = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
            L.check(lib.ia_deform_filter_compact(L.i64(P), L.i32(I), L.ptr(x), L.ptr(valid), L.ptr(cnt), L.ptr(start), L.ptr(x),
                                                 L.ptr(src), L.ptr(None), L.ptr(total), L.ptr(tmp), C.c_size_t(tmp.numel() * 8), st),
                    "ia_deform_filter_compact")
            Q = int(total.item())
            self._check_voxels()
            return x.reshape(-1, 3)[:Q], (src[:Q] if with_src else None), cnt, start, Q
        nbytes = int(lib.ia_deform_filter_tiles_tmp_bytes(L.i64(P)))
        tmp = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        L.check(lib.ia_deform_filter_tiles(L.i64(P), L.i32(I), L.ptr(x), L.ptr(valid), L.ptr(cnt), L.ptr(start), L.ptr(src), L.ptr(None),
                                           L.ptr(total), L.ptr(tmp), C.c_size_t(tmp.numel() * 8), st), "ia_deform_filter_tiles")
        Q = int(total.item())
        self._check_voxels()
        cand_x = torch.empty((Q, 3), device=dev)
        cand_src = torch.empty(Q, dtype=torch.int32, device=dev) if with_src else None
        L.check(lib.ia_deform_pack_tiles(L.i64(P), L.i32(I), L.ptr(x), L.ptr(src), L.ptr(start), L.ptr(cand_x), L.ptr(cand_src),
                                         L.ptr(tmp), st), "ia_deform_pack_tiles")
        return cand_x, cand_src, cnt, start, Q

    SPEC_ROWS = os.environ.get("IA_BROYDEN_SPEC_ROWS", "1") == "1"
    SEARCH_TOKEN = os.environ.get("IA_SEARCH_TOKEN", "0") == "1"
    SEARCH_TOKEN_MIN_POINTS = int(os.environ.get("IA_SEARCH_TOKEN_MIN_POINTS", str(1 << 22)))
    _search_lock = __import__("threading").Lock()
    _search_event = None

    @torch.no_grad()
    def _candidates(self, pts: Tensor, with_src: bool, want_fwd: bool = False, want_jinv: bool = False, order: Optional[Tensor] = None,
                    normalize=None, split: bool = False):
        """search + candidate bookkeeping for P posed points -> (cand_x [Q,3], cand_src [Q] | None, cnt [P], start [P], Q, fwd_J, J_inv).
        Default (early-filter search): the search kernel itself leaves each point's surviving candidates in its 3-slot row plus their
        count and the scan of the counts (fast_snarf.fuse_broyden_spec_rows: 44 B per point instead of 169) -- no x [P,13,3], no
        is_valid, no K9 pass; one segmented copy makes the packed list.  Points the kernel redid with the filter off (~2e-4) get their
        rows from K9 on all 13 results (rows_flagged_kernel; a 4th.. survivor as an overflow record).  spec_eps = 0: search() + _pack_candidates().
        normalize = (center [3], scale [3]): cand_x comes back as (x - center) / scale + 0.5 (the hash grid's coordinates).
        split (SDF-only queries, with_src False): the list comes back in the SPLIT layout of ia_deform_rows_pack_split -- every point's first
        candidate at first_pos[p], the others from n_first on -- and the tuple ends with (first_pos, n_first); None, None when the batch took
        another path (the caller then reads the list point-major); first_pos = (in-tile positions [P], tile offsets [ceil(P / 1024)])."""
        P, I = pts.shape[0], self.init_bones.shape[0]
        dev = self.device
        if not (self.SPEC_ROWS and self.spec_eps >= 1e-4 and P >= self.SPEC_MIN_POINTS and self.tfs.shape[0] == 1):
            if order is not None:
                pts = pts[order.long()].contiguous()
            r = self.search(pts, want_fwd=want_fwd, want_jinv=want_jinv)
            x, valid, fwd = r[0], r[1], r[2]
            J_inv = r[3] if want_jinv else None
            res = self._normalized((*self._pack_candidates(x, valid, with_src=with_src), fwd, J_inv), normalize)
            return res + (None, None) if split else res
        lib, st = L.lib(), L.stream()
        x_rows = torch.empty((P, 3, 3), device=dev)
        Jinv = torch.empty((1, P, I, 3, 3), device=dev) if want_jinv else None
        fwd = torch.empty((1, P, I, 3, 3), device=dev) if want_fwd else None
        cnt, meta, start, ovf_head = (torch.empty(P, dtype=torch.int32, device=dev) for _ in range(4))
        # overflow records + the list of points the kernel redoes with the filter off (1 / 64 of the batch): a grow-only work area
        # (locals, not attributes: concurrent calls on several streams -- render.compute_indirect_radiance -- each have their own)
        ovf_scratch = L.scratch("spec_rows", int(lib.ia_spec_rows_overflow_bytes(L.i64(P))), dev)
        ovf_cap = self._tls.ovf_cap = int(lib.ia_spec_rows_overflow_capacity(L.i64(P)))
        tot = torch.empty(2, dtype=torch.int32, device=dev)
        # IA_SEARCH_TOKEN=1 (experiment, measured without effect -- DESIGN 4.5): one search on the DEVICE at a time; this stream's search
        # starts when the previous search of any stream has finished, so that a search shares the device with another stream's gather /
        # head kernels instead of with another search
        token = self.SEARCH_TOKEN and P >= self.SEARCH_TOKEN_MIN_POINTS
        if token:
            self._search_lock.acquire()
        try:
            if token and self._search_event is not None:
                torch.cuda.current_stream(dev).wait_event(self._search_event)
            fast_snarf.fuse_broyden_spec_rows(x_rows, pts.reshape(1, P, 3), fast_snarf.ChannelLastVoxelJ(self.voxel_J_cl), self.tfs, self.init_bones,
                                              Jinv, cnt, meta, start, ovf_head, ovf_scratch, tot, self.offset_kernel, self.scale_kernel,
                                              1e-5, 1e-1, self.spec_eps, fwd_J=fwd, counters=self.spec_counters, order=order,
                                              cell_tight=self.cell_tight)
            if token:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                type(self)._search_event = ev
        finally:
            if token:
                self._search_lock.release()
        Q, n_over = tot.tolist()                                     # the one read-back of the call
        self._tls.n_over = n_over                                    # points the kernel searched again with the filter off
        self._check_voxels()
        if os.environ.get("IA_DEBUG_FLAGGED"):
            import sys
            print(f"[spec rows] P={P} Q={Q} redone={n_over} ({n_over / max(P, 1):.2e}) cap={ovf_cap}", file=sys.stderr)
        if n_over > ovf_cap:
            # more points to redo than the flagged list holds (never seen): this batch goes through is_valid + K9 instead
            del x_rows, cnt, meta, start, Jinv, fwd
            if order is not None:
                pts = pts[order.long()].contiguous()
            r = self.search(pts, want_fwd=want_fwd, want_jinv=want_jinv)
            res = self._normalized((*self._pack_candidates(r[0], r[1], with_src=with_src), r[2], (r[3] if want_jinv else None)), normalize)
            return res + (None, None) if split else res
        if self.spec_canary > 0:
            self._canary(pts, order, x_rows, cnt, meta)
        cand_x = torch.empty((Q, 3), device=dev)
        if split and not with_src and not want_fwd and not want_jinv and self.SPLIT_CANDIDATES:
            tiles = (P + 1023) // 1024
            first_pos = torch.empty(P, dtype=torch.int32, device=dev)
            tile_off_n = torch.empty(tiles + 1, dtype=torch.int32, device=dev)         # [tiles] offsets + [1] n_first
            L.check(lib.ia_deform_rows_pack_split(L.i64(P), L.i32(I), L.ptr(x_rows), L.ptr(cnt), L.ptr(meta), L.ptr(start), L.ptr(ovf_head),
                                                  L.ptr(ovf_scratch), L.ptr(first_pos), L.ptr(tile_off_n), L.ptr(tile_off_n[tiles:]), L.ptr(cand_x),
                                                  L.ptr(normalize[0].contiguous().float() if normalize else None),
                                                  L.ptr(normalize[1].contiguous().float() if normalize else None),
                                                  L.ptr(L.scan_tmp(tiles + 1, dev, extra_bytes=4 * tiles + 1024)), st), "ia_deform_rows_pack_split")
            return cand_x, None, cnt, start, Q, None, None, (first_pos, tile_off_n[:tiles]), tile_off_n[tiles:]
        cand_src = torch.empty(Q, dtype=torch.int32, device=dev) if with_src else None
        L.check(lib.ia_deform_rows_pack(L.i64(P), L.i32(I), L.ptr(x_rows), L.ptr(cnt), L.ptr(meta), L.ptr(start), L.ptr(ovf_head),
                                        L.ptr(ovf_scratch), L.ptr(cand_x), L.ptr(cand_src),
                                        L.ptr(normalize[0].contiguous().float() if normalize else None),
                                        L.ptr(normalize[1].contiguous().float() if normalize else None), st), "ia_deform_rows_pack")
        res = (cand_x, cand_src, cnt, start, Q, (fwd[0] if want_fwd else None), (Jinv[0] if want_jinv else None))
        return res + (None, None) if split else res

    # the SDF-only queries gather their candidates in the split layout (first candidates, then the rest: -6 % in the hash gather)
    SPLIT_CANDIDATES = os.environ.get("IA_SPLIT_CANDIDATES", "1") == "1"
    SPEC_CANARY = int(os.environ.get("IA_SPEC_CANARY", "0"))

    @torch.no_grad()
    def _canary(self, pts: Tensor, order: Optional[Tensor], x_rows: Tensor, cnt: Tensor, meta: Tensor):
        """runtime check of the early filter (IA_SPEC_CANARY=k, deformer.spec_canary): every k-th point of the batch is searched again
        with all 13 inits run to their end + K9 (fuse_broyden + filter: the reference's fuse_cuda_kernel_fast.cu:252-452 and
        filter.cu:10-54 semantics) and its candidate set is compared with the row the early-filter search left -- same count, same
        inits, bit-identical roots.  Accumulates on the device (canary_totals(): points checked, points that differ, points with
        overflow records -- a 4th survivor; only their count is compared); no host sync.  ~1.7 / k of the search time."""
        P, I = cnt.shape[0], self.init_bones.shape[0]
        dev = pts.device
        k = int(self.spec_canary)
        idx = torch.arange(getattr(self._tls, "canary_phase", 0) % k, P, k, device=dev)
        self._tls.canary_phase = getattr(self._tls, "canary_phase", 0) + 7          # another residue class every call
        if idx.numel() == 0:
            return
        src = order.long()[idx] if order is not None else idx
        sub = pts[src].contiguous()
        n = sub.shape[0]
        x = torch.empty((1, n, I, 3), device=dev)
        valid = torch.empty((1, n, I), dtype=torch.bool, device=dev)
        fast_snarf.fuse_broyden(x, sub.reshape(1, n, 3), None, fast_snarf.ChannelLastVoxelJ(self.voxel_J_cl), self.tfs, self.init_bones,
                                True, None, valid, self.offset_kernel, self.scale_kernel, 1e-5, 1e-1)
        keep = fast_snarf.filter(x, valid)[0]                                        # [n, I] after K9
        x = x[0]
        c, m, rows = cnt[idx].long(), meta[idx], x_rows[idx]
        flagged = m < 0                                                              # bit 31: rows + overflow records
        bad = c != keep.sum(1)
        inits = torch.arange(I, device=dev)
        want_bits = (keep.long() << inits[None]).sum(1)                              # the surviving inits as a bit set
        got_bits = torch.zeros_like(want_bits)
        for j in range(3):
            live = (c > j) & ~flagged
            init = ((m >> (8 * j)) & 0xFF).long().clamp(max=I - 1)
            got_bits = got_bits | torch.where(live, torch.ones_like(c) << init, torch.zeros_like(c))
            ref = torch.gather(x, 1, init[:, None, None].expand(n, 1, 3))[:, 0]
            same = (rows[:, j].view(torch.int32) == ref.view(torch.int32)).all(1)
            bad = bad | (live & ~same)
        bad = bad | (~flagged & (got_bits != want_bits))
        # one accumulator per STREAM (its stream orders the updates): the secondary march starts fresh worker threads on every call but
        # re-uses its side streams, so keying by thread would add a device tensor to the list per march
        skey = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        with self._canary_lock:
            acc = self._canary_by_stream.get(skey)
            if acc is None:
                acc = self._canary_by_stream[skey] = torch.zeros(3, dtype=torch.int64, device=dev)
                self._canary_parts.append(acc)
        acc += torch.stack([torch.full((), n, dtype=torch.int64, device=dev), bad.sum(), flagged.sum()])

    def canary_totals(self, reset: bool = False):
        """(points checked, points whose candidate row differs from search-to-the-end + K9, overflow points compared by count only)
        since the last reset, over all threads; synchronises the device."""
        torch.cuda.synchronize(self.device)
        with self._canary_lock:
            tot = [int(v) for v in torch.stack(self._canary_parts).sum(0).tolist()] if self._canary_parts else [0, 0, 0]
            if reset:
                for a in self._canary_parts:
                    a.zero_()
        return tuple(tot)

    @staticmethod
    def _normalized(res, normalize):
        if normalize is None:
            return res
        return ((res[0] - normalize[0]) / normalize[1] + 0.5, *res[1:])

    @torch.no_grad()
    def deform_sdf(self, pts: Tensor, geometry, order: Optional[Tensor] = None) -> Tensor:
        """SDF at posed points, nothing else: SNARFDeformer.deform with with_grad = with_feature = False as the no-grad coarse
        passes call it (coarse_alpha_fn / alpha_fn / coarse_alpha_sdf_fn).  Same search, filter and min-select as deform();
        the candidates go through geometry.sdf_only and only sdf [P] is produced (1e5 where no candidate survives).
        order (int32 permutation, optional): evaluate the points in that order (pts[order[k]] is the k-th point searched; the
        search reads through the index, no gathered copy) -- the result comes back in the caller's order."""
        pts = pts.contiguous().float()
        P, I = pts.shape[0], self.init_bones.shape[0]
        dev = self.device
        lib, st = L.lib(), L.stream()
        # the candidates leave the packing kernel in the hash grid's coordinates (three elementwise passes over [Q,3] less)
        cand_x, _, cnt, start, Q, _, _, first_pos, n_first = self._candidates(pts, with_src=False, order=order,
                                                                             normalize=(geometry.center, geometry.scale), split=True)
        csdf = geometry.sdf_only(cand_x, normalized=True)
        sdf = torch.empty(P, device=dev)
        if first_pos is not None:     # split layout: first candidates at first_pos, the others from n_first on
            L.check(lib.ia_deform_select_min_split(L.i64(P), L.ptr(start), L.ptr(cnt), L.ptr(first_pos[0]), L.ptr(first_pos[1]), L.ptr(n_first),
                                                   L.ptr(csdf), L.ptr(order), L.ptr(sdf), st), "ia_deform_select_min_split")
        elif order is not None:         # pts = caller's points[order]: the result goes back to the caller's order on the way out
            L.check(lib.ia_deform_select_min_scatter(L.i64(P), L.ptr(start), L.ptr(cnt), L.ptr(csdf), L.ptr(order), L.ptr(sdf), st),
                    "ia_deform_select_min_scatter")
        else:
            L.check(lib.ia_deform_select_min(L.i64(P), L.ptr(start), L.ptr(cnt), L.ptr(csdf), L.ptr(sdf), st), "ia_deform_select_min")
        return sdf

    @torch.no_grad()
    def deform(self, pts: Tensor, geometry, with_grad: bool = False, with_feature: bool = False, want_fwd: bool = False,
               want_jinv: bool = False):
        """SNARFDeformer.deform (snarf_deformer.py:187-261).
        returns dict(pts_cano, sdf, valid[, sdf_grad, sdf_grad_cano][, feature], + bookkeeping)."""
        pts = pts.contiguous().float()
        P, I = pts.shape[0], self.init_bones.shape[0]
        dev = self.device
        lib, st = L.lib(), L.stream()
        cand_x, cand_src, cnt, start, Q, fwd, J_inv = self._candidates(pts, with_src=True, want_fwd=with_grad or want_fwd, want_jinv=want_jinv)
        # SDF network on the packed candidates
        cg = None
        r = geometry(cand_x, with_grad=with_grad, with_feature=True)      # feature[:, 0] is the SDF
        if with_grad:
            _, cg, cf = r
        else:
            _, cf = r
        csdf, sdf_stride = cf, 13
        out = dict(pts_cano=torch.empty((P, 3), device=dev), sdf=torch.empty(P, device=dev),
                   valid=torch.empty(P, dtype=torch.bool, device=dev), sel=torch.empty(P, dtype=torch.int32, device=dev),
                   cand_src=cand_src, n_candidates=Q, fwd_J=fwd, J_inv=J_inv)
        if with_grad:
            out["sdf_grad"] = torch.empty((P, 3), device=dev)
            out["sdf_grad_cano"] = torch.empty((P, 3), device=dev)
        if with_feature:
            out["feature"] = torch.empty((P, 13), device=dev)
        L.check(lib.ia_deform_select(
            L.i64(P), L.ptr(start), L.ptr(cnt), L.ptr(cand_x), L.ptr(cand_src), L.ptr(csdf), L.i32(sdf_stride),
            L.ptr(cg), L.ptr(cf if with_feature else None), L.i32(13), L.i32(13), L.ptr(fwd),
            L.ptr(out["pts_cano"]), L.ptr(out["sdf"]), L.ptr(out["valid"]), L.ptr(out["sel"]),
            L.ptr(out.get("sdf_grad")), L.ptr(out.get("sdf_grad_cano")), L.ptr(out.get("feature")), st),
            "ia_deform_select")
        return out
