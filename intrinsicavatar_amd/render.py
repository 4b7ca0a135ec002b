"""render_step on the MI355X kernels: host-side mirror of IntrinsicAvatarModel.forward_
(models/intrinsic_avatar.py:950-1651), radiance + SDF-geometry part (BASELINE config 2):

  1  rays world -> SMPL space                     snarf_deformer.py:128-147
  2  primary march through the occupancy grid     intrinsic_avatar.py:1158-1182   -> ia_traverse_grids_*
  3  2x importance re-sampling (no grad)          :1185-1238  coarse_alpha_fn :955-998, alpha_fn :1000-1030
        SDF at edges / mid-points (deformer + SDF net) -> alpha -> T2 weights -> K2 merge(16) -> keep fg
  4  intervals -> (t_starts, t_ends, ray_indices) :1242-1247
  5  shade + composite                            :1272-1287  rendering_with_normals_sdf (volrend.py:638-807)
        deformer + SDF(grad, feature) -> normals, alpha, radiance -> T2 -> T3 (rgb, normal, opacity, depth)

Every random tensor of the reference is an explicit input: `jitter` for the
stratified near plane.  All heavy work is in libia_amd.so; torch is used for allocation and the
boolean-mask bookkeeping between the operators (exactly where the reference uses it).
"""
import os
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib as L
from . import lib_nerfacc, nerfacc
from .deformer import SNARFDeformer
from .nerfacc import RayIntervals


def ray_points(rays_o: Tensor, rays_d: Tensor, ray_indices: Tensor, t0: Tensor, t1: Optional[Tensor] = None) -> Tensor:
    n = ray_indices.shape[0]
    pts = torch.empty((n, 3), device=rays_o.device)
    L.check(L.lib().ia_ray_points(L.i64(n), L.ptr(rays_o), L.ptr(rays_d), L.ptr(ray_indices), L.ptr(t0), L.ptr(t1),
                                  L.ptr(pts), L.stream()), "ia_ray_points")
    return pts


def laplace_alpha(sdf: Tensor, dists, beta: Tensor) -> Tensor:
    """get_alpha (intrinsic_avatar.py:390-394); dists: tensor [n] or python float."""
    n = sdf.shape[0]
    alpha = torch.empty_like(sdf)
    d_t, d_c = (dists, 0.0) if isinstance(dists, Tensor) else (None, float(dists))
    L.check(L.lib().ia_laplace_alpha(L.i64(n), L.ptr(sdf), L.ptr(d_t), L.f32(d_c), L.ptr(beta), L.ptr(alpha), L.stream()),
            "ia_laplace_alpha")
    return alpha


def laplace_alpha_intervals(sdf: Tensor, t_starts: Tensor, t_ends: Tensor, beta: Tensor) -> Tensor:
    """laplace_alpha(sdf, t_ends - t_starts, beta) without materialising the difference (ia_laplace_alpha_intervals: bit-identical)."""
    alpha = torch.empty_like(sdf)
    L.check(L.lib().ia_laplace_alpha_intervals(L.i64(sdf.shape[0]), L.ptr(sdf), L.ptr(t_starts.contiguous()), L.ptr(t_ends.contiguous()), L.ptr(beta),
                                               L.ptr(alpha), L.stream()), "ia_laplace_alpha_intervals")
    return alpha


def shade_prep(sdf_grad: Tensor, rays_d: Tensor, ray_indices: Tensor, w2s_rot: Tensor):
    n = sdf_grad.shape[0]
    dev = sdf_grad.device
    ns, nw, rf = (torch.empty((n, 3), device=dev) for _ in range(3))
    L.check(L.lib().ia_shade_prep(L.i64(n), L.ptr(sdf_grad), L.ptr(rays_d), L.ptr(ray_indices), L.ptr(w2s_rot), L.ptr(ns),
                                  L.ptr(nw), L.ptr(rf), L.stream()), "ia_shade_prep")
    return ns, nw, rf


def plan_secondary_chunks(M: int, chunk: int, n_streams: int = 1, min_chunk: int = 1 << 22):
    """[(c0, c1), ...] covering rays [0, M) for compute_indirect_radiance.  One stream: chunks of `chunk` rays (the last one shorter).
    Several streams: at most 5 / 8 of `chunk` per stream -- the live working set of two chunks stays that of one serial chunk (141 against
    144 GiB on the headline step; what grows is the allocator's reserve, one pool per stream) --, EQUAL chunks (sizes differ by at most one
    ray), their number a multiple of the streams, so that the static assignment (chunk j to thread j mod n) is balanced; a batch that
    fits one chunk is split over the streams; min_chunk bounds the chunk size from below for batches of more than n_streams * min_chunk
    rays (the caller only takes the streams for large batches, SECONDARY_STREAMS_MIN_RAYS).  With fewer rays than chunks the empty
    chunks are dropped (then, and only then, the count is not a multiple of the streams)."""
    if M <= 0:
        return []
    if n_streams > 1:
        cmax = max(min(chunk * 5 // 8, -(-M // n_streams)), min_chunk, 1)
        n_chunks = n_streams * (-(-M // (n_streams * cmax)))
        bounds = [M * i // n_chunks for i in range(n_chunks + 1)]            # bounds[i] = M i / n: equal to within one ray
        return [(a, b) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    chunk = max(int(chunk), 1)
    return [(c0, min(c0 + chunk, M)) for c0 in range(0, M, chunk)]


class RenderStep:
    """One frame's render_step.  `occ_binaries` [1,64,64,64] bool + `occ_aabb` [1,6] = the (test-time)
    occupancy grid of the frame (prepare_test_occupancy_grid, intrinsic_avatar.py:360-381)."""

    def __init__(self, geometry, radiance, density, deformer: SNARFDeformer, occ_binaries: Tensor, occ_aabb: Tensor,
                 render_step_size: float, importance_sample: bool = True):
        self.geometry, self.radiance, self.density, self.deformer = geometry, radiance, density, deformer
        self.binaries, self.aabbs = occ_binaries, occ_aabb
        self._grid_bits = None
        self.render_step_size = float(render_step_size)
        self.importance_sample = importance_sample

    @property
    def grid_bits(self) -> Tensor:
        """bit-packed occupancy grid for the traversal kernel; re-packed when `binaries` is replaced or written in place."""
        b = self.binaries
        key = (b.data_ptr(), b._version, str(b.device))
        if self._grid_bits is None or self._grid_bits[0] != key:
            self._grid_bits = (key, nerfacc.pack_occupancy_bits(b[0]))
        return self._grid_bits[1]

    # ------------------------------------------------------------------ helpers
    def _beta(self) -> Tensor:
        d = self.density
        return d.beta_value() if hasattr(d, "beta_value") else d.get_beta().detach().reshape(1).float().contiguous()

    _CONST = {}

    @classmethod
    def _const(cls, value: float, n: int, dev) -> Tensor:
        """[n] float tensor filled with `value`: a view of a cached, grow-only constant (near / far planes of every march: two fill
        launches per call otherwise).  Read-only by convention."""
        key = (float(value), str(dev), torch.cuda.current_stream(dev).cuda_stream if str(dev).startswith("cuda") else 0)
        t = cls._CONST.get(key)
        if t is None or t.shape[0] < n:
            t = cls._CONST[key] = torch.full((max(n + n // 4, 4096),), float(value), device=dev)
        return t[:n]

    SORT_MIN_POINTS = 1 << 20
    # points per deformer search: P * 13 (point, init) items must stay below 2^31 (165.2 M points) and x / valid take 169 B per
    # point (25 GB at 150 M); the headline step's 16 Mi-ray secondary chunks average 145 M sample points
    MAX_SEARCH_POINTS = int(os.environ.get("IA_MAX_SEARCH_POINTS", str(150_000_000)))
    K2_MERGED_READBACK = os.environ.get("IA_K2_MERGED_READBACK", "1") == "1"

    SORT_DROP_BITS = int(os.environ.get("IA_SORT_DROP_BITS", "0"))     # low Morton bits left unsorted (0, 3 or 6)

    def _sort_grid_params(self):
        """(origin, 1 / cell) of the Morton grid of _spatial_order, cached per occupancy box (ONE attribute assignment: several host threads
        may ask at once)."""
        gkey = (self.aabbs.data_ptr(), self.aabbs._version)
        c = getattr(self, "_sort_grid_cache", None)
        if c is None or c[0] != gkey:
            # 10 bits per axis over the bounding box of the occupancy grid (every marched sample lies inside it): 2.5 mm cells for
            # a 2.5 m box.  Measured on the headline step: 1 cm cells 970 ms, 5 mm 948 ms, 2.5 mm 946 ms, 4 cm 1019 ms
            box = self.aabbs[0].tolist()
            ext = max(box[3] - box[0], box[4] - box[1], box[5] - box[2]) + 0.04
            c = (gkey, [box[0] - 0.02, box[1] - 0.02, box[2] - 0.02], float(os.environ.get("IA_SORT_INV_CELL", 1023.0 / ext)))
            self._sort_grid_cache = c
        return c[1], c[2]

    @torch.no_grad()
    def _spatial_order(self, pts: Tensor) -> Tensor:
        """int32 permutation that lists posed-space points by the Morton code of their cell (ia_morton_order)."""
        n = pts.shape[0]
        import ctypes as C
        lo, inv_cell = self._sort_grid_params()
        origin = (C.c_float * 3)(*lo)
        lib, st = L.lib(), L.stream()
        order = torch.empty(n, dtype=torch.int32, device=pts.device)
        nb = int(lib.ia_morton_order_tmp_bytes(L.i64(n)))
        tmp = L.scratch("morton", nb, pts.device)                             # 256-byte aligned (a fresh allocation's base)
        L.check(lib.ia_morton_order(L.i64(n), L.ptr(pts), origin, L.f32(inv_cell), L.i32(self.SORT_DROP_BITS), L.ptr(order), L.ptr(tmp),
                                    C.c_size_t(nb), st), "ia_morton_order")
        return order

    @torch.no_grad()
    def _sdf_at(self, pts: Tensor) -> Tensor:
        """SDF of posed-space points (deformer search + SDF network, min over the candidates).  Large batches are evaluated
        in SPATIAL order (Morton code of the point's cell, 1024 cells per axis over the grid's box): the searches of neighbouring
        points walk the same voxels of the skinning grid and their candidates share hash-grid cells, so both gather kernels run
        out of the vector L1 instead of L2 (profiles/r02_broyden_probe.json); the values are those of the
        unsorted evaluation, only the schedule changes."""
        n = pts.shape[0]
        if n > self.MAX_SEARCH_POINTS:
            # the search keeps x [P,13,3] + valid [P,13] (169 B / point) and its packing needs P * 13 < 2^31: bound the batch in
            # POINTS, whatever the caller's ray chunk produced (a dense occupancy grid gives 64 samples per secondary ray)
            return torch.cat([self._sdf_at(pts[a:a + self.MAX_SEARCH_POINTS]) for a in range(0, n, self.MAX_SEARCH_POINTS)])
        if n < self.SORT_MIN_POINTS or os.environ.get("IA_SORT_POINTS", "1") != "1":
            return self.deformer.deform_sdf(pts, self.geometry)
        # the search reads the points THROUGH the permutation and the min-select writes through it: no gathered copy of the points
        # (the 12-byte random gather dragged 64-byte sectors: 4.7 x its algorithmic bytes, profiles/r02_pmc_traffic.json)
        return self.deformer.deform_sdf(pts, self.geometry, order=self._spatial_order(pts))

    # ------------------------------------------------------------------ sampling (no grad)
    @torch.no_grad()
    def sample(self, rays: Tensor, jitter: Optional[Tensor] = None):
        """steps 1-4 of forward_: world->SMPL rays, primary march, 2x importance resampling.
        returns (rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats)."""
        dfm = self.deformer
        rays = dfm.transform_rays_w2s(rays.float())
        n_rays = rays.shape[0]
        rays_o, rays_d, far = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous(), rays[:, 7]      # (two 12-byte-row copies: the kernels take [n,3])
        beta = self._beta()
        # -- 2. primary march (sampling_override, intrinsic_avatar.py:49-141; near 0 / far 1e10)
        near_planes = self._const(0.0, n_rays, rays.device)
        far_planes = self._const(1e10, n_rays, rays.device)
        if jitter is not None:
            near_planes = near_planes + jitter * self.render_step_size
        intervals, samples, _ = nerfacc.traverse_grids(rays_o, rays_d, self.binaries, self.aabbs, near_planes, far_planes,
                                                       self.render_step_size, 0.0, grid_bits=self.grid_bits, termination_planes=False)
        stats = dict(n_edges0=intervals.vals.shape[0], n_samples0=samples.vals.shape[0])
        # -- 3. importance resampling.  Host syncs are kept to the data-dependent sizes: boolean-mask indexing (one
        # nonzero + sync per use in the reference) is replaced by ONE index list per edge set, and the pairing
        # "k-th left edge <-> k-th right edge" by "right edge = left edge + 1" (consecutive samples share edges).
        smp_next = None
        if self.importance_sample and samples.vals.numel() > 0:
            for it in range(2):
                vals = intervals.vals
                if it == 0:        # coarse_alpha_fn: SDF at every edge, interval sdf = min(left, right)
                    pts = ray_points(rays_o, rays_d, intervals.ray_indices, vals)
                    sdf = self._sdf_at(pts)
                    sdf_merge = torch.empty_like(sdf)          # is_left ? min(sdf, next sdf) : 1e10, one launch
                    L.check(L.lib().ia_edge_min_sdf(L.i64(sdf.shape[0]), L.ptr(sdf), L.ptr(intervals.is_left), L.ptr(sdf_merge), L.stream()),
                            "ia_edge_min_sdf")
                    alphas = laplace_alpha(sdf_merge, self.render_step_size, beta)
                else:              # alpha_fn: SDF at interval mid-points
                    smp = smp_next            # (came back with K2's result: one read-back for the edge count and the sample count)
                    pts = ray_points(rays_o, rays_d, smp.ray_indices, smp.t_starts, smp.t_ends)
                    sdf_curr = self._sdf_at(pts)
                    alphas = laplace_alpha(smp.to_edges(sdf_curr, 1e10), smp.to_edges(smp.t_ends - smp.t_starts, 0.0), beta)
                weights, _ = nerfacc.render_weight_from_alpha(alphas, packed_info=intervals.packed_info)
                # K2 + the selection of its reached edges (intrinsic_avatar.py:1211-1226) as count -> scan -> fill kernels
                if self.K2_MERGED_READBACK:
                    rvals, ril, rir, ray_idx, pinfo, smp_next = lib_nerfacc.ray_resampling_merge_compact_samples(
                        intervals.packed_info, vals, intervals.is_left, intervals.is_right, weights, 16)
                else:           # IA_K2_MERGED_READBACK=0 (A/B hook): K2's edge count and the sample count in two read-backs
                    rvals, ril, rir, ray_idx, pinfo = lib_nerfacc.ray_resampling_merge_compact(
                        intervals.packed_info, vals, intervals.is_left, intervals.is_right, weights, 16)
                    smp_next = lib_nerfacc.interval_samples(pinfo, rvals, ril, ray_idx)
                intervals = RayIntervals(vals=rvals, is_left=ril, is_right=rir, ray_indices=ray_idx, packed_info=pinfo)
        # -- 4.
        smp = smp_next if smp_next is not None else \
            lib_nerfacc.interval_samples(intervals.packed_info, intervals.vals, intervals.is_left, intervals.ray_indices)
        t_starts, t_ends, ray_indices, packed_info = smp.t_starts, smp.t_ends, smp.ray_indices, smp.packed_info
        stats["n_samples"] = t_starts.shape[0]
        return rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats

    # ------------------------------------------------------------------ forward (eval)
    @torch.no_grad()
    def forward(self, rays: Tensor, jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
        dfm = self.deformer
        rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats = self.sample(rays, jitter)
        beta = self._beta()
        # -- 5. shade + composite (rgb_normal_alpha_fn + rendering_with_normals_sdf)
        pts = ray_points(rays_o, rays_d, ray_indices, t_starts, t_ends)
        d = dfm.deform(pts, self.geometry, with_grad=True, with_feature=True)
        w2s_rot = dfm.w2s[:3, :3].contiguous()
        normal_smpl, normal_world, refl01 = shade_prep(d["sdf_grad"], rays_d, ray_indices, w2s_rot)
        dists = t_ends - t_starts
        alphas = laplace_alpha(d["sdf"], dists, beta)
        rgbs = self.radiance(d["pts_cano"], d["feature"], refl01, normal_world)
        weights, trans = nerfacc.render_weight_from_alpha(alphas, packed_info=packed_info)
        acc = lambda v: nerfacc._Accumulate.apply(weights, v, ray_indices, packed_info)      # noqa: E731
        colors, normals, opac = acc(rgbs), acc(normal_world), acc(None)
        depths = acc(((t_starts + t_ends) / 2.0)[:, None])
        depths = depths + (1.0 - opac) * far[:, None]                   # intrinsic_avatar.py:1287
        stats["n_candidates"] = d["n_candidates"]
        return dict(comp_rgb=colors, comp_normal=normals, opacity=opac, depth=depths, weights=weights, trans=trans,
                    alphas=alphas, rgbs=rgbs, sdf=d["sdf"], sdf_grad=d["sdf_grad"], normals=normal_smpl,
                    positions=d["pts_cano"], valid=d["valid"], t_starts=t_starts, t_ends=t_ends,
                    ray_indices=ray_indices, packed_info=packed_info, stats=stats)

    # ------------------------------------------------------------------ forward + backward (training step)
    def parameters(self):
        """trainable parameters (the empty checkpoint-compatibility entry of the SH encoding is not one)."""
        ps = list(self.geometry.parameters()) + list(self.radiance.parameters()) + list(self.density.parameters())
        return [p for p in ps if p.requires_grad]

    def forward_backward(self, rays: Tensor, target_rgb: Tensor, target_mask: Optional[Tensor] = None,
                         jitter: Optional[Tensor] = None, curv_u: Optional[Tensor] = None, lambda_curv: float = 0.0,
                         loss_scale: float = 1.0, **loss_kw) -> Dict[str, Tensor]:
        """one optimisation step's fwd+bwd (training_step, systems/intrinsic_avatar.py:160-251, rgb/eikonal/mask
        terms): no-grad sampling, differentiable shading + compositing, loss, backward to every parameter."""
        from . import train
        rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats = self.sample(rays, jitter)
        if curv_u is not None:
            curv_u = curv_u[:t_starts.shape[0]]
        out = train.shade_differentiable(self, rays_o, rays_d, ray_indices, t_starts, t_ends, packed_info, curv_u=curv_u)
        loss = train.training_loss(out, target_rgb, target_mask, lambda_curv=lambda_curv, **loss_kw)
        (loss * loss_scale if loss_scale != 1.0 else loss).backward()
        out["loss"] = loss.detach()
        out["stats"] = stats
        return out

    def forward_backward_phys(self, rays: Tensor, target_rgb: Tensor, material, emitter, spp: int, light_u: Tensor,
                              shuffle_u: Tensor, target_mask: Optional[Tensor] = None, jitter: Optional[Tensor] = None,
                              render_mode: str = "uniform_light", env_base: Optional[Tensor] = None,
                              background_color: Optional[Tensor] = None, global_illumination: bool = False,
                              light_sampling: str = "shared", loss_scale: float = 1.0, retain_graph: bool = False,
                              material_jitter: Optional[Tensor] = None, loss_config: Optional[dict] = None,
                              eik_denominator=None) -> Dict[str, Tensor]:
        """BASELINE config 4: training step with the PBR branch (material head, volume scattering, secondary rays,
        light / uniform_light estimator) -- fwd + bwd to geometry, radiance, material and environment-light parameters.
        loss_scale: weight of this ray chunk when a frame is processed in several chunks with gradient accumulation
        (n_chunk_rays / n_frame_rays makes the accumulated gradient that of the frame-mean loss); retain_graph: keep the
        graph of `env_base` (one generated environment image shared by all chunks of a step).
        material_jitter [n_samples,3] ~ N(0,1): the material jitter pass (:1116-1140) for the smoothness maps.
        loss_config: None = training_loss_phys (L1 rgb + eikonal + mask BCE + L1 rgb_phys on the linear maps); a dict of the
        reference's `system.loss` weights (configs/config.yaml:87-109: lambda_rgb_l1, lambda_rgb_phys_l1, lambda_mask_bce, ...) = the
        loss of IntrinsicAvatarSystem.training_step (systems/intrinsic_avatar.py:160-301) on the reference's output dict
        (train_phys.reference_training_loss; target_rgb / target_mask are the batch's `rgb` / `alpha`); its terms come back as
        out["loss_terms"].  tests/test_gpu_backward_golden.py holds this call to the reference's own autograd.
        eik_denominator: None = the eikonal mean is over this call's samples; a number, or a callable n_samples -> number (e.g. an
        all-reduce of the ranks' sample counts divided by the world size), = the denominator of that mean under ray-batch sharding, so that
        the average of the ranks' losses is the loss of the global batch (systems/intrinsic_avatar.py:235-239 takes .mean() over all samples)."""
        from . import train_phys
        rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats = self.sample(rays, jitter)
        out = train_phys.shade_differentiable_phys(self, material, emitter, rays_o, rays_d, ray_indices, t_starts, t_ends,
                                                   packed_info, spp, light_u, shuffle_u, render_mode=render_mode,
                                                   env_base=env_base, background_color=background_color,
                                                   global_illumination=global_illumination, light_sampling=light_sampling,
                                                   jitter_n=material_jitter)
        if loss_config is None:
            den = eik_denominator(int(t_starts.shape[0])) if callable(eik_denominator) else eik_denominator
            loss = train_phys.training_loss_phys(out, target_rgb, target_mask, **({} if den is None else dict(eik_denominator=den)))
        else:
            if background_color is None:
                background_color = torch.ones(3, device=rays.device)
            d = self._training_dict(out, rays.shape[0], far, t_starts, t_ends, ray_indices, packed_info, background_color, render_mode)
            loss, out["loss_terms"] = train_phys.reference_training_loss(d, target_rgb, target_mask, loss_config, material)
            out["output_dict"] = d
        (loss * loss_scale if loss_scale != 1.0 else loss).backward(retain_graph=retain_graph)
        out["loss"] = loss.detach()
        out["stats"].update(stats)
        return out

    # ------------------------------------------------------------------ secondary rays (compute_indirect_radiance)
    @torch.no_grad()
    def compute_indirect_radiance(self, rays_o: Tensor, rays_d: Tensor, near: float = 0.0, far: float = 1.5,
                                  n_secondary: int = 64, chunk: int = int(os.environ.get("IA_SECONDARY_CHUNK", str(1 << 24)))):
        """models/intrinsic_avatar.py:396-545 (eval): march each secondary ray over [near, far] (step (far-near)/63),
        SDF at the sample starts, zero-crossing resampling to 4 intervals (K4), shade them, composite.
        returns (transmittance [M,1], indirect rgb [M,3]).  Rays are processed in chunks of at most `chunk` rays (16 Mi: ~145 M sample
        points and ~25 GB of search outputs per chunk on the headline scene -- sized for 288 GB of HBM; measured 1067 -> 1008 ms per
        headline step against 2 Mi); the sample points of a chunk go through the search in batches of <= MAX_SEARCH_POINTS."""
        M = rays_o.shape[0]
        dev = rays_o.device
        hook = getattr(self._march_hooks, "enter", None)          # frame pipelining (train_phys.forward_backward_phys_pipelined): this
        if hook is not None:                                      # thread's half-frame reaches its secondary march
            hook()
        tr = torch.ones((M, 1), device=dev)
        rgb = L.zeros((M, 3), dev)
        step = (far - near) / (n_secondary - 1)
        beta = self._beta()
        w2s_rot = self.deformer.w2s[:3, :3].contiguous()
        # Chunks are independent (rays are): with SECONDARY_STREAMS > 1 they are worked off by that many host threads, each on its own
        # HIP stream, so that the kernels of two chunks share the device -- the search (vector-memory path + VALU), the hash gather
        # (L1 misses in flight) and the SDF head (matrix pipe) are bound by different units, and every chunk has data-dependent size
        # read-backs during which its stream would otherwise leave the GPU to the tail of one kernel.  Two PROCESSES on one GPU reach
        # 1.16 x the throughput of one on the headline step (tools/runs/r04_two_processes_one_gpu.sh); inside one process two streams
        # get 4 % (340.7 -> 326.5 ms per step, same box), a third stream nothing more (tools/runs/r04_streams_ab.sh); starting the second
        # thread only when the first one's first search has run (so that the streams sit in different phases of a chunk) changes nothing
        # (314.1 / 315.1 against 315.0 / 312.4 ms).  Results do not depend on the chunking (ray-batch sharding invariance),
        # so they are bit-identical to the serial loop (tests/test_gpu_relight_oracle.py).
        n_streams = self._secondary_streams_for(M, chunk, dev)
        # ray chunks (plan_secondary_chunks); a chunk whose march produced more sample points than four search batches is split
        # in two and marched again (the march costs ~1 ms): the working set is bounded in SAMPLES, not only in rays
        work = plan_secondary_chunks(M, chunk, n_streams, self.SECONDARY_MIN_CHUNK)[::-1]
        args = (rays_o, rays_d, near, far, step, beta, w2s_rot, tr, rgb)
        if n_streams <= 1 or len(work) <= 1:
            self._secondary_chunks(work, *args)
            return tr, rgb
        import threading
        _ = self.grid_bits, self._sort_grid_params()            # lazily cached host-side state: made before the threads start
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_side_streams", None) is None or len(self._side_streams) != n_streams or self._side_streams[0].device != dev:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        errors = []
        # STATIC assignment, chunk j to thread j mod n: every thread sees the same chunk sizes step after step, so its stream's allocator
        # pool holds the right blocks after one warm-up (with a shared queue the thread that happened to get the short last chunk in the
        # warm-up paid a multi-GB hipMalloc later: 0.245 ... 0.405 s per relit frame, run to run)
        chunks = work[::-1]
        works = [chunks[k::n_streams][::-1] for k in range(n_streams)]

        def worker(side, my_work):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(side), torch.no_grad():      # device, stream and grad mode are per thread
                    self._secondary_chunks(my_work, *args)
            except BaseException as e:                                                       # noqa: B902 -- handled by the caller
                errors.append(e)
        threads = []
        for side, my_work in zip(self._side_streams, works):
            side.wait_stream(main)                              # inputs were produced on the caller's stream
            threads.append(threading.Thread(target=worker, args=(side, my_work), daemon=True))
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for side in self._side_streams:
            main.wait_stream(side)                              # tr / rgb are read on the caller's stream
        if not errors:
            # what the side streams' pools hold now is what the next march can re-use: reserved - (what the caller's pool has live + cached)
            # is not exposed per pool, so the working-set estimate of this march stands in for it
            self._side_pool_bytes = n_streams * max(b - a for a, b in chunks) * self.SECONDARY_BYTES_PER_RAY
        if errors:
            if all(isinstance(e, torch.cuda.OutOfMemoryError) for e in errors):
                # the side streams' allocator pools did not fit after all (another process on the GPU, a smaller device): give their
                # blocks back and march the batch again on the caller's stream, one chunk at a time -- same results, chunking-invariant
                del errors[:]
                self._side_streams = None
                torch.cuda.synchronize(dev)
                torch.cuda.empty_cache()
                self.secondary_stream_fallbacks = getattr(self, "secondary_stream_fallbacks", 0) + 1
                # the gate let this batch through and it did not fit: do not try again on the next calls (each failed attempt costs a
                # full synchronise, an empty_cache and a second march); the streams come back after STREAM_FALLBACK_COOLDOWN marches
                self._stream_cooldown = self.STREAM_FALLBACK_COOLDOWN
                tr.fill_(1.0)
                rgb.zero_()
                self._secondary_chunks(plan_secondary_chunks(M, chunk, 1)[::-1], *args)
                return tr, rgb
            raise errors[0]
        return tr, rgb

    import threading as _threading
    _march_hooks = _threading.local()          # per host thread: .enter (callable) / .streams (override of SECONDARY_STREAMS)
    SECONDARY_STREAMS = int(os.environ.get("IA_SECONDARY_STREAMS", "2"))
    STREAM_FALLBACK_COOLDOWN = int(os.environ.get("IA_STREAM_FALLBACK_COOLDOWN", "64"))
    # bytes of device memory a marched ray of a chunk keeps live at the peak of its chunk on the headline scene (9 samples per ray on
    # average x (search outputs + sorted copies + level-major hash features)): 142 GiB live for 2 x 10.5 Mi rays = ~7 KB per ray
    SECONDARY_BYTES_PER_RAY = int(os.environ.get("IA_SECONDARY_BYTES_PER_RAY", "7168"))

    def _secondary_streams_for(self, M: int, chunk: int, dev) -> int:
        """how many HIP streams / host threads the secondary march of M rays takes.  Each side stream has its own caching-allocator pool
        (measured: 184 GiB reserved against 142 GiB live on the headline step, for 3-4 % of the step), so the streams are only taken
        when the device has room for that: free + this process's cached-but-unused memory >= 1.3 x the working set of the chunks in
        flight (the measured reserve / live ratio); otherwise the serial loop runs (IA_SECONDARY_STREAMS=1 forces it: 99 GiB live /
        113 GiB reserved on the headline step, 3 % slower).  An OOM inside the threads falls back to it too."""
        n = getattr(self._march_hooks, "streams", None) or self.SECONDARY_STREAMS
        if n <= 1 or dev.type != "cuda" or M <= self.SECONDARY_STREAMS_MIN_RAYS:
            self.last_secondary_streams = 1
            return 1
        if getattr(self, "_stream_cooldown", 0) > 0:          # an earlier march ran out of memory on the side streams
            self._stream_cooldown -= 1
            self.last_secondary_streams = 1
            return 1
        plan = plan_secondary_chunks(M, chunk, n, self.SECONDARY_MIN_CHUNK)
        in_flight = n * max(b - a for a, b in plan) * self.SECONDARY_BYTES_PER_RAY
        free, _total = torch.cuda.mem_get_info(dev)
        # usable by the side streams: device-free memory plus the blocks already cached in THEIR pools (steady state: the previous step's
        # working set sits there).  Blocks cached in the caller's pool are not available to another stream, so they do not count.  The
        # allocator's snapshot (segments carry their stream; milliseconds) is only taken when neither free memory alone nor free memory plus the
        # working set the side streams held after their last successful march settles it.
        need = 1.3 * in_flight
        have_side = getattr(self, "_side_streams", None) is not None
        if free < need and not (have_side and free + getattr(self, "_side_pool_bytes", 0) >= need):
            # (steady state is settled by the working-set estimate of the last successful march -- no allocator call per step; a wrong
            #  estimate costs one OOM fallback and the cool-down)
            side_ids = {st.cuda_stream for st in (getattr(self, "_side_streams", None) or [])}
            side = 0
            if side_ids:
                try:
                    for seg in torch.cuda.memory_snapshot():
                        if seg.get("device", dev.index) == dev.index and seg.get("stream") in side_ids:
                            side += seg["total_size"] - seg["allocated_size"]
                except Exception:               # noqa: BLE001 -- a diagnostic API: fall back to the working-set estimate of the last march
                    side = getattr(self, "_side_pool_bytes", 0)
            n = n if free + side >= need else 1
        self.last_secondary_streams = n
        return n
    # below ~8 M rays the kernels of a chunk no longer fill the device and splitting them makes it worse (config-4 shape, 1 M secondary rays:
    # 21.8 ms per step serial, 24.2 on two streams)
    SECONDARY_STREAMS_MIN_RAYS = int(os.environ.get("IA_SECONDARY_STREAMS_MIN_RAYS", str(1 << 23)))
    SECONDARY_MIN_CHUNK = int(os.environ.get("IA_SECONDARY_MIN_CHUNK", str(1 << 22)))

    def _secondary_chunks(self, work, rays_o, rays_d, near, far, step, beta, w2s_rot, tr, rgb):
        """works the chunks of `work` (this thread's own list, last one first) into tr / rgb."""
        dev = rays_o.device
        while work:
            c0, c1 = work.pop()
            ro, rd = rays_o[c0:c1].contiguous(), rays_d[c0:c1].contiguous()
            m = ro.shape[0]
            intervals, samples, _ = nerfacc.traverse_grids(
                ro, rd, self.binaries, self.aabbs, self._const(near, m, dev), self._const(far, m, dev),
                step, 0.0, grid_bits=self.grid_bits, max_extent=far - near, incoherent=True, termination_planes=False)
            if samples.vals.shape[0] > 4 * self.MAX_SEARCH_POINTS and m > 1:
                del intervals, samples
                work.extend([(c0 + m // 2, c1), (c0, c0 + m // 2)])
                continue
            t_starts, t_ends = samples.interval_ends(intervals)
            ray_indices = samples.ray_indices
            if t_starts.numel() == 0:
                continue
            # coarse_alpha_sdf_fn (:399-428): SDF at the interval STARTS
            sdf = self._sdf_at(ray_points(ro, rd, ray_indices, t_starts))
            alphas = laplace_alpha_intervals(sdf, t_starts, t_ends, beta)
            pinfo = lib_nerfacc.pack_info(ray_indices, m)
            # (capacity-sized outputs: compact_foreground walks them through rpi, the size read-back of cdf.cu:511 is not needed)
            rpi, rs, re, is_fg = lib_nerfacc.ray_resampling_sdf_fine(pinfo, t_starts[:, None], t_ends[:, None], alphas, sdf, 4, exact_size=False)
            # keep the foreground intervals (:516-528): count -> scan over rays -> segmented copy, and the packed_info of the kept list
            ray_indices, t_starts, t_ends, pinfo = lib_nerfacc.compact_foreground(rpi, rs, re, is_fg)
            if t_starts.numel() == 0:
                continue                                   # no zero crossing anywhere: fully transmissive
            # rendering(rgb_alpha_fn) (:430-456, volrend.py:19-194)
            pts = ray_points(ro, rd, ray_indices, t_starts, t_ends)
            np_ = pts.shape[0]
            if np_ >= self.SORT_MIN_POINTS and os.environ.get("IA_SORT_POINTS", "1") == "1":
                # the shading points of a chunk in spatial order as well (search, hash gathers with Jacobian, both field heads;
                # 1.5 ns -> 1.1 ns per point in the search): per-point work is order-free, only alpha and rgb go back to ray order
                order = self._spatial_order(pts)
                ps = torch.empty_like(pts)
                L.check(L.lib().ia_gather_rows3_i32(L.i64(np_), L.ptr(pts), L.ptr(order), L.ptr(ps), L.stream()), "ia_gather_rows3_i32")
                ol = order.long()
                ri_s, dt_s = ray_indices[ol], (t_ends - t_starts)[ol]
                d = self.deformer.deform(ps, self.geometry, with_grad=True, with_feature=True)
                _, normal_world, refl01 = shade_prep(d["sdf_grad"], rd, ri_s, w2s_rot)
                a_s = laplace_alpha(d["sdf"], dt_s, beta)
                rgbs_s = self.radiance(d["pts_cano"], d["feature"], refl01, normal_world).contiguous()
                a, rgbs = torch.empty_like(a_s), torch.empty_like(rgbs_s)
                L.check(L.lib().ia_scatter_f32_i32(L.i64(np_), L.ptr(a_s), L.ptr(order), L.ptr(a), L.stream()), "ia_scatter_f32_i32")
                L.check(L.lib().ia_scatter_rows3_i32(L.i64(np_), L.ptr(rgbs_s), L.ptr(order), L.ptr(rgbs), L.stream()), "ia_scatter_rows3_i32")
            else:
                d = self.deformer.deform(pts, self.geometry, with_grad=True, with_feature=True)
                _, normal_world, refl01 = shade_prep(d["sdf_grad"], rd, ray_indices, w2s_rot)
                a = laplace_alpha_intervals(d["sdf"], t_starts, t_ends, beta)
                rgbs = self.radiance(d["pts_cano"], d["feature"], refl01, normal_world)
            w, _ = nerfacc.render_weight_from_alpha(a, packed_info=pinfo)
            acc = nerfacc._Accumulate.apply(w, None, ray_indices, pinfo)
            torch.neg(acc, out=tr[c0:c0 + m])                     # tr = 1 - acc written in place: (-acc) + 1 is the same IEEE operation
            tr[c0:c0 + m].add_(1.0)
            rgb[c0:c0 + m] = nerfacc._Accumulate.apply(w, rgbs, ray_indices, pinfo)

    # ------------------------------------------------------------------ relighting (render_mode = light)
    @torch.no_grad()
    def relight(self, rays: Tensor, material, emitter, spp: int, light_u: Tensor, shuffle_u: Tensor,
                background_color: Optional[Tensor] = None, global_illumination: bool = False,
                jitter: Optional[Tensor] = None, render_mode: str = "light", scatter_u: Optional[Tensor] = None,
                return_index_lists: bool = False) -> Dict[str, Tensor]:
        """forward_ with enable_phys and render_mode='light' (BASELINE configs 3 / 5):
        rendering_with_normals_mats_sdf (volrend.py:810-1020) -> sample_volume_interaction (pbr/utils.py:70-229)
        -> per-ray shuffled light directions (:1356-1378) -> secondary rays (:396-545) -> pbr_light_forward (:755-861)
        -> accumulate(resampled_weights, Lo) (:1427-1466).  light_u [spp,3], shuffle_u [n_rays,spp]: explicit RNG."""
        from . import pbr
        dfm = self.deformer
        rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats = self.sample(rays, jitter)
        n_rays = rays_o.shape[0]
        dev = rays_o.device
        beta = self._beta()
        w2s_rot = dfm.w2s[:3, :3].contiguous()
        if background_color is None:
            background_color = torch.ones(3, device=dev)
        # -- shade with materials (rgb_normal_mats_alpha_fn, :1066-1156)
        pts = ray_points(rays_o, rays_d, ray_indices, t_starts, t_ends)
        d = dfm.deform(pts, self.geometry, with_grad=True, with_feature=True)
        normal_smpl, normal_world, refl01 = shade_prep(d["sdf_grad"], rays_d, ray_indices, w2s_rot)
        alphas = laplace_alpha_intervals(d["sdf"], t_starts, t_ends, beta)
        rgbs, enc2, xp2 = self.radiance(d["pts_cano"], d["feature"], refl01, normal_world, return_embedding=True)
        mats = material(enc2, xp2, d["feature"], self.radiance.prog.mask(self.radiance.global_step, dev))
        weights, trans = nerfacc.render_weight_from_alpha(alphas, packed_info=packed_info)
        acc = lambda v: nerfacc._Accumulate.apply(weights, v, ray_indices, packed_info)      # noqa: E731
        out = dict(comp_rgb=acc(rgbs), comp_normal=acc(normal_world), albedo=acc(mats[:, :3].contiguous()),
                   roughness=acc(mats[:, 3:4].contiguous()), metallic=acc(mats[:, 4:5].contiguous()), opacity=acc(None))
        out["depth"] = acc(((t_starts + t_ends) / 2.0)[:, None]) + (1.0 - out["opacity"]) * far[:, None]
        out["packed_info"] = packed_info
        # (references to the per-sample tensors of the primary samples, as forward() returns them: no copy)
        out["primary_samples"] = dict(sdf_grad=d["sdf_grad"], valid=d["valid"], t_starts=t_starts, t_ends=t_ends, ray_indices=ray_indices)
        extras = dict(weights=weights, sdf=d["sdf"], alphas=alphas, normals=normal_smpl, albedo=mats[:, :3],
                      roughness=mats[:, 3:4], metallic=mats[:, 4:5])
        rgb_phys = background_color[None].expand(n_rays, 3).clone()
        demod_phys = rgb_phys.clone()
        if render_mode == "uniform_light":
            out["visibility"] = torch.zeros((n_rays, 1), device=dev)                      # :1266-1267
        stats.update(n_resampled=0, n_fg=0, n_secondary=0)
        if ray_indices.numel() > 0:
            # -- volume-interaction re-sampling (sample_volume_interaction, models/pbr/utils.py:70-229): K1 + layout scans
            vi = pbr.VolumeInteraction(ray_indices, t_starts, t_ends, n_rays, spp, weights, d["sdf"])
            stats["n_resampled"], stats["n_fg"] = vi.R, vi.F
            transmittance = 1.0 - out["opacity"]
            if vi.F > 0:
                F_ = vi.F
                w_fg, nrm, alb, rough, metal = vi.gather(rays_o, rays_d, weights, normal_smpl, mats[:, :3], mats[:, 3:4], mats[:, 4:5])
                pos, view = vi.positions, vi.view_dirs
                ind = lambda c: c if global_illumination else None      # noqa: E731
                if render_mode in ("light", "uniform_light"):
                    if render_mode == "light":
                        # light directions: sampled once per frame (prepare, :292-305), rotated to SMPL space
                        # (transform_dirs_w2s), permuted per ray (:1356-1378)
                        dirs_smpl = emitter.sample(spp, light_u, w2s_rot=w2s_rot)
                        inv_pdf_all = None
                    else:
                        # stratified uniform sphere (:680-689); directions are used as-is in SMPL space
                        assert spp == 512, "uniform_light asserts samples_per_pixel == 512 (:1392)"
                        dirs_smpl, inv_pdf_all = pbr.uniform_sphere_stratified(16, 32, light_u[:, :2])
                    shuffled = vi.shuffle(shuffle_u)
                    ro, rd, src, out_dirs = pbr.secondary_rays(nrm, pos, dirs_smpl, dir_index=shuffled)
                    inv_pdf = inv_pdf_all[shuffled.long()] if render_mode == "uniform_light" else None
                    stats["n_secondary"] = int(ro.shape[0])
                    t_, c_ = self.compute_indirect_radiance(ro, rd)
                    sec_tr, sec_rgb = pbr.scatter_secondary(F_, src, t_, c_)
                    res = pbr.pbr_shade(render_mode, nrm, alb, rough, metal, view, out_dirs, sec_tr, ind(sec_rgb), emitter, w2s_rot,
                                        inv_pdf=inv_pdf)
                    fg_Lo, fg_demod = res[0], res[1] + res[2]
                    out["shuffled"] = shuffled
                    if render_mode == "uniform_light":
                        zero3 = torch.zeros(3, device=dev)
                        out["visibility"] = vi.composite(w_fg, res[3], torch.zeros_like(transmittance), zero3).mean(-1, keepdim=True)
                elif render_mode in ("mats", "mis"):
                    # scatterer.sample (+ emitter.sample per point for mis), :547-652 / :863-948; explicit uniforms scatter_u
                    assert scatter_u is not None and scatter_u.shape[0] >= F_, "mats / mis need scatter_u [n_fg, 6]"
                    sc_dirs = pbr.brdf_sample(nrm, view, rough, scatter_u[:F_, :3])
                    if render_mode == "mis":
                        li_dirs = emitter.sample(F_, scatter_u[:F_, 3:6].contiguous(), w2s_rot=w2s_rot)
                        out_dirs = torch.cat([sc_dirs, li_dirs], 0)
                        rep = lambda t: t.repeat(2, 1)      # noqa: E731
                    else:
                        out_dirs, rep = sc_dirs, (lambda t: t)
                    stats["n_secondary"] = int(out_dirs.shape[0])
                    sec_tr, sec_rgb = self.compute_indirect_radiance(rep(pos).contiguous(), out_dirs.contiguous())
                    Lo2, Ld2, Ls2 = pbr.pbr_shade(render_mode, rep(nrm), rep(alb), rep(rough), rep(metal), rep(view), out_dirs, sec_tr,
                                                  ind(sec_rgb), emitter, w2s_rot)
                    fg_Lo = Lo2.reshape(2, F_, 3).sum(0) if render_mode == "mis" else Lo2
                    fg_demod = (Ld2 + Ls2).reshape(2, F_, 3).sum(0) if render_mode == "mis" else Ld2 + Ls2
                else:
                    raise NotImplementedError(f"Render mode {render_mode} not supported.")
                # Lo.scatter_(bg_indices, background) + accumulate_along_rays(resampled_weights, Lo) (:1335-1342,:1420-1466)
                rgb_phys = vi.composite(w_fg, fg_Lo, transmittance, background_color)
                demod_phys = vi.composite(w_fg, fg_demod.contiguous(), transmittance, background_color)     # Lo_demod = Lo_diff + Lo_spec (:1421-1436)
                out["secondary_tr"], out["fg_Lo"] = sec_tr, fg_Lo
                if render_mode in ("light", "uniform_light"):
                    out["secondary_rgb"] = sec_rgb
                out["fg_extras"] = dict(positions=pos, normals=nrm, albedo=alb, roughness=rough, metallic=metal, t_dirs=view)
            out["resampled_packed_info"] = vi.resampled_packed_info
            out["sampled_indices"] = vi.sampled_idx            # source interval of every re-sample (K1's `indices`): a reference, no launch
            if return_index_lists:
                fg_idx, bg_idx, rri, rw = vi.index_lists(weights, transmittance)
                out.update(resampled_ray_indices=rri, resampled_weights=rw, fg_indices=fg_idx, bg_indices=bg_idx)
        out.update(comp_rgb_phys=rgb_phys, comp_demod_phys=demod_phys, stats=stats)
        return out

    # ------------------------------------------------------------------ the reference's output dict
    @staticmethod
    def output_dict(o: Dict[str, Tensor], background_color: Tensor, render_mode: str, n_samples: int) -> Dict[str, Tensor]:
        """the dict IntrinsicAvatarModel.forward_ returns with enable_phys (models/intrinsic_avatar.py:1492-1651): the linear
        maps, the constant-background dict (`*_bg`) and the composited sRGB dict (`*_full`), same keys / shapes / dtypes."""
        from . import pbr
        acc = o["opacity"]
        n, dev = acc.shape[0], acc.device
        bgc = background_color.to(dev).float()
        out = dict(comp_rgb=o["comp_rgb"], comp_normal=o["comp_normal"], opacity=acc, depth=o["depth"], rays_valid=acc > 0,
                   rays_valid_phys=acc > 0, num_samples=torch.as_tensor([n_samples], dtype=torch.int32, device=dev),
                   comp_rgb_phys=o["comp_rgb_phys"], comp_demod_phys=o["comp_demod_phys"], comp_albedo=o["albedo"],
                   comp_metallic=o["metallic"], comp_roughness=o["roughness"])
        if render_mode == "uniform_light":
            out["visibility"] = o["visibility"]
        for k in ("sdf_samples", "sdf_grad_samples", "sdf_laplace_samples", "weights", "points", "intervals", "ray_indices",
                  "normals_orientation_loss_map", "albedo_smoothness_loss_map", "roughness_smoothness_loss_map", "metallic_smoothness_loss_map"):
            if k in o:
                out[k] = o[k]                                                              # training form (:1519-1597)
        bgm = bgc.mean().reshape(1, 1).expand(n, 1)
        out_bg = dict(comp_rgb=bgc[None].expand(n, 3), num_samples=torch.zeros_like(out["num_samples"]),
                      rays_valid=torch.zeros_like(out["rays_valid"]), rays_valid_phys=torch.zeros_like(out["rays_valid_phys"]),
                      comp_albedo=torch.zeros((n, 3), device=dev), comp_metallic=bgm, comp_roughness=bgm)
        T = 1.0 - acc
        out_full = dict(comp_rgb=pbr.rgb_to_srgb(out["comp_rgb"] + out_bg["comp_rgb"] * T).clamp(0, 1),
                        num_samples=out["num_samples"] + out_bg["num_samples"], rays_valid=out["rays_valid"] | out_bg["rays_valid"],
                        rays_valid_phys=out["rays_valid_phys"] | out_bg["rays_valid_phys"],
                        comp_rgb_phys=pbr.rgb_to_srgb(out["comp_rgb_phys"]).clamp(0, 1),
                        comp_demod_phys=pbr.rgb_to_srgb(out["comp_demod_phys"]).clamp(0, 1),
                        comp_albedo=out["comp_albedo"] + out_bg["comp_albedo"] * T,
                        comp_metallic=out["comp_metallic"] + out_bg["comp_metallic"] * T,
                        comp_roughness=out["comp_roughness"] + out_bg["comp_roughness"] * T)
        return {**out, **{k + "_bg": v for k, v in out_bg.items()}, **{k + "_full": v for k, v in out_full.items()}}

    @torch.no_grad()
    def forward_(self, rays: Tensor, material, emitter, spp: int, light_u: Tensor, shuffle_u: Optional[Tensor] = None,
                 background_color: Optional[Tensor] = None, global_illumination: bool = False, render_mode: str = "light",
                 scatter_u: Optional[Tensor] = None, jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """IntrinsicAvatarModel.forward_ in eval mode with enable_phys (models/intrinsic_avatar.py:950-1651): relight() + the
        reference's output dict.  Random tensors are explicit."""
        if background_color is None:
            background_color = torch.ones(3, device=rays.device)
        o = self.relight(rays, material, emitter, spp, light_u, shuffle_u, background_color=background_color,
                         global_illumination=global_illumination, jitter=jitter, render_mode=render_mode, scatter_u=scatter_u)
        return self.output_dict(o, background_color, render_mode, o["stats"]["n_samples"])

    def forward_train_(self, rays: Tensor, material, emitter, spp: int, light_u: Optional[Tensor], jitter: Optional[Tensor] = None,
                       material_jitter: Optional[Tensor] = None, background_color: Optional[Tensor] = None,
                       global_illumination: bool = False, render_mode: str = "light", shuffle_u: Optional[Tensor] = None,
                       env_base: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """forward_ in train() mode (differentiable; the caller takes losses of the returned maps): stratified near plane
        (`jitter`), material jitter pass (`material_jitter` ~ N(0,1), :1116-1140), per-point light directions for render_mode
        'light' (:772-781), and the training keys of the output dict (:1519-1597)."""
        from . import train_phys
        if background_color is None:
            background_color = torch.ones(3, device=rays.device)
        rays_o, rays_d, far, t_starts, t_ends, ray_indices, packed_info, stats = self.sample(rays, jitter)
        res = train_phys.shade_differentiable_phys(
            self, material, emitter, rays_o, rays_d, ray_indices, t_starts, t_ends, packed_info, spp, light_u, shuffle_u,
            render_mode=render_mode, env_base=env_base, background_color=background_color, global_illumination=global_illumination,
            jitter_n=material_jitter, light_sampling="per_point" if render_mode == "light" else "shared")
        d = self._training_dict(res, rays.shape[0], far, t_starts, t_ends, ray_indices, packed_info, background_color, render_mode)
        d["stats"] = dict(res["stats"], **stats)
        return d

    def _training_dict(self, res, n_rays, far, t_starts, t_ends, ray_indices, packed_info, background_color, render_mode):
        """the output dict of forward_ in train() mode (:1492-1651) from shade_differentiable_phys's result."""
        dev = far.device
        mid = (t_starts + t_ends) / 2.0
        depth = nerfacc._Accumulate.apply(res["weights"].detach(), mid[:, None].contiguous(), ray_indices, packed_info)
        T = 1.0 - res["opacity"]
        o = dict(res, depth=depth + T.detach() * far[:, None], comp_demod_phys=res["comp_rgb_phys"])
        if "volume_interaction" in res:
            o["comp_demod_phys"] = res["volume_interaction"].composite(res["fg_weights"], (res["fg_Lo_diff"] + res["fg_Lo_spec"]).contiguous(),
                                                                       T, background_color)
        z1 = torch.zeros((n_rays, 1), device=dev)
        o.update(sdf_samples=res["sdf"], sdf_grad_samples=res["sdf_grad"], sdf_laplace_samples=torch.zeros_like(res["sdf"]),
                 points=mid, intervals=t_ends - t_starts, ray_indices=ray_indices)
        for k in ("normals_orientation_loss_map", "albedo_smoothness_loss_map", "roughness_smoothness_loss_map", "metallic_smoothness_loss_map"):
            o.setdefault(k, z1)
        if render_mode == "uniform_light":
            if "volume_interaction" in res and "visibility" not in o:
                # vis = 2 * secondary transmittance per foreground re-sample (pbr_uniform_light_forward :741), composited with the
                # re-sampling weights, no background (:1427-1470); a monitoring map: no gradient, built only for the output dict
                with torch.no_grad():
                    vis3 = (2.0 * res["secondary_tr"]).expand(-1, 3).contiguous()
                    o["visibility"] = res["volume_interaction"].composite(res["fg_weights"].detach(), vis3, torch.zeros_like(T),
                                                                          torch.zeros(3, device=dev)).mean(-1, keepdim=True)
            o.setdefault("visibility", z1)
        return self.output_dict(o, background_color, render_mode, int(t_starts.shape[0]))
