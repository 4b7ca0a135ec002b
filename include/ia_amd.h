/*
 * ia_amd.h -- C ABI of libia_amd.so: the MI355X (gfx950) implementation of
 * IntrinsicAvatar's volumetric render_step hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *     is enqueued on it, nothing synchronises unless stated;
 *   - the caller owns every buffer; data-dependent output sizes use a two-phase
 *     "count -> (caller allocates) -> fill" protocol;
 *   - return value: IA_OK (0) or a negative IA_ERR_*; ia_last_error() gives text;
 *   - bool tensors are 1 byte per element (torch.bool layout);
 *   - tensors are contiguous, row-major, in the shapes the reference's Python
 *     operator passes (cited per entry point as file:line under /root/reference).
 *
 * The reference has no C plugin ABI: its boundary is the Python operator surface
 * of five modules.  Each entry point below names the reference
 * operator it replaces; INTEGRATION.md shows the ctypes binding that maps the
 * reference's call sites onto this ABI.
 */
#ifndef IA_AMD_H
#define IA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IA_OK 0
#define IA_ERR_INVALID (-1)     /* bad argument (the reference would TORCH_CHECK) */
#define IA_ERR_LAUNCH (-2)      /* HIP launch / runtime error */
#define IA_ERR_UNSUPPORTED (-3) /* valid for the reference API but not on this path */

typedef void* ia_stream_t;

int ia_version(void);
const char* ia_last_error(void);

/* ------------------------------------------------------------------------- */
/* Prefix sums (plumbing for the two-phase protocol).                          */
/* out[i] = sum_{j<i} in[j]; total (1 element, device) = sum of all; tmp must  */
/* hold ia_scan_tmp_bytes(n) bytes.                                            */
int64_t ia_scan_tmp_bytes(int64_t n);
int ia_exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t* total, int64_t n, void* tmp, ia_stream_t stream);
int ia_exclusive_scan_i32(const int32_t* in, int32_t* out, int32_t* total, int64_t n, void* tmp, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* nerfacc.traverse_grids  (nerfacc==0.5.3; reference call sites
 * models/occ_grid/temporal_occ_grid.py:166-175, models/intrinsic_avatar.py:84-93).
 * One grid level (the reference always passes `binaries[t_idx:t_idx+1]`).
 *
 * ia_occgrid_pack_bits: binaries bool[rx*ry*rz] -> bit-packed uint32[(rx*ry*rz+31)/32]
 * (bit c&31 of word c>>5 is cell c, c = (x*ry + y)*rz + z).
 *
 * Phase 1 (count): one DDA walk per ray; per-ray counts packed as  n_edges | (n_samples << 32)  in one int64 (so a
 *                  single ia_exclusive_scan_i64 serves both) and per-ray run descriptors into `scratch`
 *                  (ia_traverse_scratch_bytes(n_rays) bytes).
 * Phase 2 (fill):  caller passes the exclusive scan of the packed counts and zero-filled flag arrays; expands the
 *                  run descriptors into RayIntervals{vals,is_left,is_right,ray_indices,packed_info[n,2] i64} and
 *                  RaySamples{vals,ray_indices,packed_info[n,2] i64} and termination planes.
 * Rays are independent; output order = ray order, then marching order.  step_size must be > 0.
 */
int ia_occgrid_pack_bits(const uint8_t* binaries, int64_t n_cells, uint32_t* bits, ia_stream_t stream);

int64_t ia_traverse_scratch_bytes(int64_t n_rays);

int ia_traverse_grids_count(
    int64_t n_rays, const float* rays_o /*[n,3]*/, const float* rays_d /*[n,3]*/,
    const uint32_t* grid_bits, int res_x, int res_y, int res_z, const float* aabb /*[6] device*/,
    const float* near_planes /*[n]*/, const float* far_planes /*[n]*/, float step_size, float cone_angle,
    void* scratch, int64_t* packed_counts /*[n]*/, ia_stream_t stream);

int ia_traverse_grids_fill(
    int64_t n_rays, const float* rays_o, const float* rays_d,
    const uint32_t* grid_bits, int res_x, int res_y, int res_z, const float* aabb,
    const float* near_planes, const float* far_planes, float step_size, float cone_angle,
    const void* scratch, const int64_t* packed_counts /*[n]*/, const int64_t* packed_starts /*[n]*/,
    int64_t* iv_packed_info /*[n,2] or NULL*/, int64_t* sm_packed_info /*[n,2] or NULL*/,
    float* iv_vals /*[E]*/, uint8_t* iv_is_left /*[E] zeroed*/, uint8_t* iv_is_right /*[E] zeroed*/,
    int64_t* iv_ray_indices /*[E]*/, float* sm_vals /*[S]*/, int64_t* sm_ray_indices /*[S]*/,
    float* termination_planes /*[n] or NULL*/, ia_stream_t stream);

/* Single-launch variant for callers that can bound the output size (count + look-back scan + coalesced fill in one
 * kernel; identical outputs).  cap_edges / cap_samples = element capacity of the output arrays (< 2^31); totals[3]
 * (device) receives {n_edges, n_samples, overflow}: when overflow != 0 the outputs are incomplete and the caller must
 * fall back to the two-phase protocol above.  The flag arrays need NOT be zeroed.  sm_t_starts / sm_t_ends (both or
 * neither): the interval ends per SAMPLE, i.e. iv_vals[iv_is_left] / iv_vals[iv_is_right] as every caller of traverse_grids
 * forms them next (occ_grid sampling, models/intrinsic_avatar.py:396-428), without the two boolean-mask gathers.  scratch:
 * ia_traverse_fused_scratch_bytes(n_rays) bytes.  termination_planes [n_rays] or NULL: with NULL a ray's walk ends where it leaves
 * the cell box of the OCCUPIED cells (nothing is emitted beyond it; the plane -- t after the last, empty cells -- is the only output
 * that needs the rest of the walk); intervals / samples are identical either way. */
int64_t ia_traverse_fused_scratch_bytes(int64_t n_rays);
int ia_traverse_grids_fused(int64_t n_rays, const float* rays_o, const float* rays_d, const uint32_t* grid_bits,
                            int rx, int ry, int rz, const float* aabb, const float* near_planes,
                            const float* far_planes, float step_size, float cone_angle, void* scratch,
                            int64_t cap_edges, int64_t cap_samples, int64_t* totals, int64_t* iv_packed_info,
                            int64_t* sm_packed_info, float* iv_vals, uint8_t* iv_is_left, uint8_t* iv_is_right,
                            int64_t* iv_ray_indices, float* sm_vals, int64_t* sm_ray_indices,
                            float* termination_planes, float* sm_t_starts /*[cap_samples] or NULL*/,
                            float* sm_t_ends /*[cap_samples] or NULL*/, int span_sorted /* != 0: walk each 1024-ray tile in order of the rays' box-crossing span (incoherent rays); outputs identical.  n > 1: n = upper bound of the samples of one ray (the sort's bins; with 1 it is cap_samples / n_rays -- capacities may be sized for the average ray) */,
                            ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* nerfacc.render_weight_from_alpha / accumulate_along_rays
 * (call sites models/intrinsic_avatar.py:506,1199,1427-1453; models/volrend.py:162,176-187,764,783-797).
 * packed_info is int32 [n_rays,2] = (start, count), samples sorted by ray.        */
int ia_render_weight_from_alpha(int64_t n_rays, const int32_t* packed_info, const float* alphas,
                                float* weights, float* trans, ia_stream_t stream);
int ia_render_weight_from_alpha_bwd(int64_t n_rays, const int32_t* packed_info, const float* alphas,
                                    const float* weights, const float* trans,
                                    const float* g_weights /*or NULL*/, const float* g_trans /*or NULL*/,
                                    float* g_alphas, ia_stream_t stream);
/* out[r, :] = sum_{i in ray r} w_i * v_i   (values NULL => dim 1, v = 1), in sample order */
int ia_accumulate_along_rays(int64_t n_rays, const int32_t* packed_info, int dim, const float* weights,
                             const float* values /*[S,dim] or NULL*/, float* out /*[n,dim]*/, ia_stream_t stream);
/* backward: g_w[i] = <g_out[r], v_i>,  g_v[i,:] = w_i g_out[r]  (either output may be NULL) */
int ia_accumulate_along_rays_bwd(int64_t n_samples, int dim, const int64_t* ray_indices, const float* weights,
                                 const float* values, const float* g_out, float* g_weights, float* g_values,
                                 ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* lib.nerfacc pack / unpack  (lib/nerfacc/pack.py:12-190, cuda/csrc/pack.cu:7-164) */
int ia_pack_info(int64_t n_samples, const int64_t* ray_indices, int64_t n_rays,
                 int32_t* packed_info /*[n,2]*/, void* tmp /* ia_scan_tmp_bytes(n_rays)+4*n_rays+8 bytes */,
                 ia_stream_t stream);
int ia_unpack_info(int64_t n_rays, const int32_t* packed_info, int64_t* ray_indices, ia_stream_t stream);
int ia_unpack_info_to_mask(int64_t n_rays, const int32_t* packed_info, int n_samples,
                           uint8_t* masks /*[n,n_samples] zeroed*/, ia_stream_t stream);
int ia_unpack_data(int64_t n_rays, const int32_t* packed_info, int data_dim, const float* data,
                   int n_samples_per_ray, float* out /*[n,S,D] zeroed*/, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* lib.nerfacc importance resampling (lib/nerfacc/cdf.py:13-244, cuda/csrc/cdf.cu).
 * Prologue (cdf.cu:177-183): resample_packed_info[r] = (excl. cumsum, cnt),
 *   cnt = (steps>0 ? n : 0) + (add_steps ? steps : 0); total -> 1 device int32.  */
int ia_resample_packed_info(int64_t n_rays, const int32_t* packed_info, int n, int add_steps,
                            int32_t* resample_packed_info, int32_t* total,
                            void* tmp /* ia_scan_tmp_bytes(n_rays)+4*n_rays bytes */, ia_stream_t stream);
/* K1..K4 share one design (csrc/resample_math.h): the launch's sample positions u_j are ONE table, phase A leaves each ray's CDF as a
 * table (one lane per ray: the only serial recurrences), phase B inverts it per OUTPUT element, 64 consecutive elements per store.
 * n_in = number of input intervals (K2: edges), n_out = resample total of ia_resample_packed_info, tmp = ia_resample_tmp_bytes(n_rays,
 * n_in, n) bytes of scratch.  EVERY output element is written by the kernels (the reference's launchers zero / -1 initialise them,
 * cdf.cu:189-191,372-377,512-514: not needed here). */
size_t ia_resample_tmp_bytes(int64_t n_rays, int64_t n_in, int n);
/* K1 ray_resampling (cdf.cu:10-215) */
int ia_ray_resampling(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts, const float* ends,
                      const float* weights, const float* sdfs, const int32_t* resample_packed_info, int64_t n_out,
                      float* resample_ts, float* resample_offsets, int64_t* surface_idx,
                      int64_t* resample_indices, int32_t* resample_fg_counts, int32_t* resample_bg_counts, void* tmp,
                      ia_stream_t stream);
/* K1 with capacity-sized outputs: n_out = n x n_rays slots, the true total (ia_resample_packed_info's *total = n x rays with
 * samples; cdf.cu:183's `.item()`) read on the DEVICE through n_out_dev (NULL: ia_ray_resampling); the slots behind it stay unwritten.  The caller
 * takes the total together with its next size (pbr.VolumeInteraction: with the foreground count) and hands on [:total] views. */
int ia_ray_resampling_upto(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts, const float* ends,
                           const float* weights, const float* sdfs, const int32_t* resample_packed_info, int64_t n_out,
                           const int32_t* n_out_dev, float* resample_ts, float* resample_offsets, int64_t* surface_idx,
                           int64_t* resample_indices, int32_t* resample_fg_counts, int32_t* resample_bg_counts, void* tmp,
                           ia_stream_t stream);
/* K2 ray_resampling_merge (cdf.cu:217-401) */
int ia_ray_resampling_merge(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* vals,
                            const uint8_t* is_left, const uint8_t* is_right, const float* weights,
                            const int32_t* resample_packed_info, float* resample_vals, float* resample_dists,
                            uint8_t* resample_is_left, uint8_t* resample_is_right, uint8_t* is_resample,
                            uint8_t* is_fg_sample, void* tmp, ia_stream_t stream);
/* K2 with its caller's compaction fused in (models/intrinsic_avatar.py:1221-1226: the edges with is_fg are kept -- nonzero, four
 * boolean-mask gathers, unpack_info, pack_info): ..._count leaves cnt [n_rays] (kept edges per ray), start (exclusive scan) and
 * *total (the one size read-back); ..._fill writes vals / is_left / is_right / ray_indices [total] and packed_info [n_rays,2] of the
 * kept edges straight to their final places.  tmp (ia_resample_tmp_bytes) carries the tables between the two calls; scan_tmp:
 * ia_scan_tmp_bytes(n_rays). */
int ia_ray_resampling_merge_count(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* vals,
                                  const uint8_t* is_left, const uint8_t* is_right, const float* weights, int32_t* cnt,
                                  int32_t* start, int32_t* total, void* tmp, void* scan_tmp, ia_stream_t stream);
int ia_ray_resampling_merge_fill(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* vals,
                                 const uint8_t* is_left, const uint8_t* is_right, const int32_t* cnt, const int32_t* start,
                                 float* out_vals, uint8_t* out_is_left, uint8_t* out_is_right, int64_t* out_ray_indices,
                                 int32_t* out_packed_info, void* tmp, ia_stream_t stream);
/* K3 ray_resampling_fine (cdf.cu:403-534); tmp may be NULL for n <= 8 (the points of a ray stay in registers) */
int ia_ray_resampling_fine(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts, const float* ends,
                           const float* weights, const int32_t* resample_packed_info, int64_t n_out, float* resample_starts,
                           float* resample_ends, uint8_t* is_fg_sample, void* tmp, ia_stream_t stream);
/* K4 ray_resampling_sdf_fine (cdf.cu:536-696); tmp as K3 */
int ia_ray_resampling_sdf_fine(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts,
                               const float* ends, const float* alphas, const float* sdfs,
                               const int32_t* resample_packed_info, int64_t n_out, float* resample_starts,
                               float* resample_ends, uint8_t* is_fg_sample, void* tmp, ia_stream_t stream);

/* Foreground compaction of a fine re-sampling (the caller side of K3 / K4: models/intrinsic_avatar.py:516-528 keeps the
 * intervals with is_fg -- three boolean-mask gathers + unpack_info -- and packs the kept ray indices again).  A ray's re-samples
 * are consecutive: ia_fg_count: cnt [n_rays] = kept intervals per ray, start = exclusive scan, *total = F (scan_tmp:
 * ia_scan_tmp_bytes(n_rays)); ia_fg_compact: ray_indices i64 [F], t_starts / t_ends [F] in ray order, out_packed_info i32
 * [n_rays,2] = pack_info(ray_indices, n_rays). */
int ia_fg_count(int64_t n_rays, const int32_t* resampled_packed_info, const uint8_t* is_fg, int32_t* cnt, int32_t* start, int32_t* total,
                void* scan_tmp, ia_stream_t stream);
int ia_fg_compact(int64_t n_rays, const int32_t* resampled_packed_info, const uint8_t* is_fg, const float* starts, const float* ends,
                  const int32_t* cnt, const int32_t* start, int64_t* ray_indices, float* t_starts, float* t_ends, int32_t* out_packed_info,
                  ia_stream_t stream);

/* The samples of an interval list -- what forward_ forms from RayIntervals after every re-sampling (models/intrinsic_avatar.py:1242-1247:
 * t_starts = vals[is_left], t_ends = vals[is_right], ray_indices[is_left], pack_info; and alpha_fn :1000-1030) -- as flag -> scan -> fill
 * over the EDGES, one size read-back: ia_interval_samples_count: pos i32 [n_edges] = exclusive scan of is_left, *total = S (scan_tmp:
 * ia_scan_tmp_bytes(n_edges)); ia_interval_samples_fill: left_idx i64 [S] (or NULL) = indices of the left edges, t_starts / t_ends [S]
 * (right edge = left edge + 1), sample_ray_indices i64 [S], sample_packed_info i32 [n_rays,2] = pack_info(sample_ray_indices, n_rays).
 * ia_samples_to_edges: out[e] = is_left[e] ? sample_vals[pos[e]] : fill  (`x = fill; x[is_left] = sample_vals`, :1022-1025). */
int ia_interval_samples_count(int64_t n_edges, const uint8_t* is_left, int32_t* pos, int32_t* total, void* scan_tmp, ia_stream_t stream);
/* ... over a CAPACITY-sized edge list whose length is still on the device (the kept edges of K2 are written into a buffer of
 * n_in + n * n_rays >= T slots, cdf.cu:370's `.item()` is not taken; *n_edges = T): pos i32 [capacity], slots behind the list count nothing, *total = S
 * -- so that T and S come back in ONE read-back (lib_nerfacc.ray_resampling_merge_compact_samples).  scan_tmp: ia_scan_tmp_bytes(capacity). */
int ia_interval_samples_count_upto(int64_t capacity, const uint8_t* is_left, const int32_t* n_edges, int32_t* pos, int32_t* total,
                                   void* scan_tmp, ia_stream_t stream);
int ia_interval_samples_fill(int64_t n_rays, int64_t n_edges, const int32_t* edge_packed_info, const float* vals,
                             const int64_t* ray_indices, const uint8_t* is_left, const int32_t* pos, const int32_t* total,
                             int64_t* left_idx, float* t_starts, float* t_ends, int64_t* sample_ray_indices,
                             int32_t* sample_packed_info, ia_stream_t stream);
int ia_samples_to_edges(int64_t n_edges, const uint8_t* is_left, const int32_t* pos, const float* sample_vals, float fill, float* out,
                        ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* fast-SNARF deformer kernels (models/deformers/fast_snarf/deformer_torch.py:86-125,
 * cuda/precompute/precompute.cu:24-103, cuda/fuse_kernel/fuse_cuda_kernel_fast.cu:250-452,
 * cuda/filter/filter.cu:10-77).
 * voxel_J layouts: IA_LAYOUT_NCDHW = the reference's [B,12,D,H,W];
 *                  IA_LAYOUT_NDHWC = channel-last [B,D,H,W,12] (48 B per voxel, the
 *                  MI355X-native layout: one trilinear corner pair = 96 contiguous bytes). */
#define IA_LAYOUT_NCDHW 0
#define IA_LAYOUT_NDHWC 1
int ia_precompute(int B, int D, int H, int W, const float* voxel_w /*[1,24,D,H,W]*/, const float* tfs /*[B,24,4,4]*/,
                  const float* offset /*[3]*/, const float* scale /*[3]*/,
                  float* voxel_d /*[B,3,D,H,W] or NULL*/, float* voxel_J /*[B,12,D,H,W] or NULL*/,
                  float* voxel_J_cl /*[B,D,H,W,12] or NULL*/, ia_stream_t stream);
/* x, J_inv, is_valid are caller-zeroed (deformer_torch.py:113-115) */
int ia_fuse_broyden(int B, int64_t N, int I, const float* xd_tgt /*[B,N,3]*/, const float* voxel_J, int layout,
                    int D, int H, int W, const float* tfs, const int32_t* bone_ids /*[I]*/,
                    const float* offset, const float* scale, float cvg_threshold, float dvg_threshold,
                    float* x /*[B,N,I,3]*/, float* J_inv /*[B,N,I,3,3] or NULL (not needed when use_j_inv=false)*/, uint8_t* is_valid /*[B,N,I]*/,
                    float* fwd_J /*[B,N,I,3,3] or NULL: forward LBS Jacobian at each root (= fwd_tfs, deformer_torch.py:49-52)*/,
                    ia_stream_t stream);
int ia_filter(int64_t N, int I, const float* x /*[1,N,I,3]*/, const uint8_t* mask, uint8_t* out, ia_stream_t stream);
/* ia_fuse_broyden with a K9-CONSISTENT EARLY FILTER (product path of every query batch; B = 1, channel-last grid;
 * no reference counterpart -- the reference runs every search to its end and lets filter.cu:10-54 drop the duplicates).
 * One lane owns one point and searches its inits in REVERSE order; a search whose next position comes within `eps`
 * (inf-norm, canonical metres) of a TIGHT root (|J_inv|_F <= 2.5) that a LATER init of the same point converged to, inside
 * that root's voxel cell, while its own |J_inv|_F <= 3, is retired (is_valid = 0): it would end within K9's 1e-4 of that
 * root, or fail -- K9 drops it either way.  A completed root between 1e-4 and 2e-4 of a recorded one (or a 4th distinct
 * root) makes the lane search its point again with the filter off.  Items that are not retired are computed with the operation
 * sequence of ia_fuse_broyden (bit-identical x / J_inv / fwd_J / is_valid), and ia_filter of the result equals ia_filter of
 * ia_fuse_broyden's on all but ~1e-7 of the points (1 of 16.4 M on the headline frame, profiles/r04_spec_search_probe.json;
 * DESIGN.md 4.5, tests/test_gpu_spec_search.py); eps = 0 retires nothing.
 * cell_tight: NULL or uint8 [D,H,W] from ia_cell_tightness: a root in a voxel cell where the TRUE Jacobian of the skinning map is not
 * tight everywhere (a fold of the map nearby: two roots 1e-4 ... 1e-3 apart that both look tight to Broyden's estimate) retires
 * nothing.  With the table the differences above vanish: 0 of 145 M points on the eight reference poses
 * (profiles/r04_spec_search_probe_poses.jsonl); without it the rule is round-4a's.
 * counters: NULL or uint64[5], caller-zeroed, accumulated: fetches issued, retired items, completed valid items, points
 * searched again with the filter off, in-range corner loads of the fetches. */
/* per voxel cell (entry = the cell whose LOW corner is the voxel; the last index of an axis holds 0): bit 0 set iff at 27 sample points
 * of the cell det(dg/dx) keeps one sign and |(dg/dx)^-1|_F <= tau, dg/dx = A(x) + sum_c dw_c/dx (A_c x + b_c) the true Jacobian of
 * g(x) = A(x) x + b(x) - xd on the trilinear voxel_J (weight-gradient term included); bit 2: that sign is positive; bit 1: the cell's
 * 26 neighbours are tight too, with the same sign (the map is coherently oriented on the neighbourhood: the retirement box of a root
 * there is not cut to its cell).  Once per pose, next to ia_precompute. */
int ia_cell_tightness(int D, int H, int W, const float* voxel_J_cl /*[D,H,W,12]*/, const float* offset, const float* scale, float tau,
                      uint8_t* cell_tight /*[D,H,W]*/, ia_stream_t stream);
int ia_fuse_broyden_spec(int64_t N, int I, const float* xd_tgt /*[N,3]*/, const float* voxel_J_cl /*[D,H,W,12]*/, int D, int H, int W,
                         const float* tfs /*[24,4,4]*/, const int32_t* bone_ids, const float* offset, const float* scale,
                         float cvg_threshold, float dvg_threshold, float eps, float* x /*[N,I,3]*/, float* J_inv /*[N,I,3,3] or NULL*/,
                         uint8_t* is_valid /*[N,I]*/, float* fwd_J /*[N,I,3,3] or NULL*/, uint64_t* counters,
                         const uint8_t* cell_tight /* NULL or [D,H,W] */, ia_stream_t stream);
/* The same search with the CANDIDATE BOOKKEEPING done in the kernel (replaces is_valid + filter.cu:10-54 + the mask indexing of
 * snarf_deformer.py:187-196 on the product path).  A search that completes valid is compared with the point's recorded roots:
 * within K9's 1e-4 it is a duplicate, from 2e-4 up it is a candidate and gets the next of the point's ia_spec_rows_slots() (= 3)
 * row slots; in between, or with the row full, the point is searched again with the filter off and K9 runs literally on its 13
 * results.  Outputs: x_rows [N, 3, 3]: the k-th candidate of a point (k = 0: its highest init) in slot k; cnt [N]; meta [N]: the
 * inits of slots 0..2 in bytes 0..2 (bit 31: the point has overflow records: a 4th, 5th ... survivor of a redone point, chained
 * per point through ovf_head [N], which is written for such points only); start [N] = exclusive scan of cnt; ovf_scratch:
 * ia_spec_rows_overflow_bytes(N) bytes.  total_and_overflow [2]: [0] = Q, [1] = number of points redone -- above
 * ia_spec_rows_overflow_capacity(N) results were lost: redo the batch with ia_fuse_broyden_spec + K9.
 * 44 bytes per point leave the kernel instead of 169 (x [N,I,3] + is_valid).  J_inv / fwd_J (optional) are written at
 * [point, init] as in ia_fuse_broyden.  scan_tmp: ia_scan_tmp_bytes(N) bytes.
 * SMALL batches (N <= IA_BR_SMALL_MAX, default 2^18 = the list's minimum capacity; the reference's 4096-ray training batches,
 * configs/sampler/edge.yaml:2): one lane per (point, init) search, every search run to its end (fuse_cuda_kernel_fast.cu:252-452 as
 * written), all N points through the list of redone points and K9 applied literally (filter.cu:10-54) -- the 13 searches of a point
 * side by side instead of one after the other in one lane; same outputs ([1] = 0 then).
 * ia_deform_rows_pack: cand_x [Q,3] (+ cand_src [Q] = point * I + init, optional) in (point, ascending init) order; with
 * norm_center / norm_scale the candidates leave as the hash grid's unit-cube coordinates (x - center) / scale + 0.5 (the three
 * elementwise passes of models/rf/geometry.py:155 `(points - self.center) / self.scale + 0.5` done on the way out; same IEEE
 * operations, same bits). */
int ia_spec_rows_slots(void);
size_t ia_spec_rows_overflow_bytes(int64_t N);        /* scratch of a call on N points: overflow records + the list of redone points (N / 64, at least 2^18) */
int64_t ia_spec_rows_overflow_capacity(int64_t N);   /* points of a call on N points that can be redone */
int ia_fuse_broyden_spec_rows(int64_t N, int I, const float* xd_tgt, const float* voxel_J_cl, int D, int H, int W, const float* tfs,
                              const int32_t* bone_ids, const float* offset, const float* scale, float cvg_threshold,
                              float dvg_threshold, float eps, float* x_rows, float* J_inv, float* fwd_J, int32_t* cnt, uint32_t* meta,
                              int32_t* start, int32_t* ovf_head, void* ovf_scratch, int32_t* total_and_overflow, void* scan_tmp,
                              uint64_t* counters, const int32_t* order /* NULL or [N]: point p = xd_tgt[order[p]] */,
                              const uint8_t* cell_tight /* NULL or [D,H,W] (ia_cell_tightness) */, ia_stream_t stream);
int ia_deform_rows_pack(int64_t N, int I, const float* x_rows, const int32_t* cnt, const uint32_t* meta, const int32_t* start,
                        const int32_t* ovf_head, const void* ovf_scratch, float* cand_x, int32_t* cand_src,
                        const float* norm_center /* NULL or [3] (device) */, const float* norm_scale /* NULL or [3] (device) */,
                        ia_stream_t stream);
/* the same in the SPLIT layout for the SDF-only queries: every point's first candidate (lowest init) at first_pos[p] + first_tile_off[p / 1024]
 * (out: first_pos [N] = exclusive count of points that have candidates inside the point's tile of 1024, first_tile_off [ceil(N / 1024)] = the
 * tiles' offsets; n_first [1], DEVICE: the number of such points), its other candidates from n_first on, point-major -- the two sub-lists are
 * each spatially coherent in canonical space, which the hash gather that follows needs (a second candidate lies on another body part).
 * scan_tmp: ia_scan_tmp_bytes(N / 1024 + 1) + 4 (N / 1024 + 1) + 512 bytes.  ia_deform_select_min_split = ia_deform_select_min[_scatter] over
 * that list (order may be NULL). */
int ia_deform_rows_pack_split(int64_t N, int I, const float* x_rows, const int32_t* cnt, const uint32_t* meta, const int32_t* start,
                              const int32_t* ovf_head, const void* ovf_scratch, int32_t* first_pos, int32_t* first_tile_off, int32_t* n_first,
                              float* cand_x, const float* norm_center, const float* norm_scale, void* scan_tmp, ia_stream_t stream);
int ia_deform_select_min_split(int64_t P, const int32_t* start, const int32_t* cnt, const int32_t* first_pos, const int32_t* first_tile_off,
                               const int32_t* n_first, const float* cand_sdf, const int32_t* order, float* sdf, ia_stream_t stream);
/* diagnostics (no reference counterpart): runs the searches of ia_fuse_broyden without outputs and ACCUMULATES into
 * counters[17] (caller-zeroed): [0] trilinear fetches, [1] in-range corner loads, [2] converged & in-box items,
 * [3] diverged, [4] out of iterations, [5+k] items that ended after k fetches (k = 2..11).  Used by bench.py to price the
 * search against the vector-memory (L1) path. */
int ia_broyden_stats(int B, int64_t N, int I, const float* xd_tgt, const float* voxel_J, int layout, int D, int H, int W,
                     const float* tfs, const int32_t* bone_ids, const float* offset, const float* scale,
                     float cvg_threshold, float dvg_threshold, uint64_t* counters, ia_stream_t stream);
/* diagnostics: how many distinct voxels the 64 lanes of a wave touch in the first two fetches of every search, with the items in
 * point-major (init_major = 0) or init-major order; counters [16] caller-zeroed (layout: broyden_voxel_stats_kernel, snarf.hip) */
int ia_broyden_voxel_stats(int64_t N, int I, int init_major, const float* xd_tgt, const float* voxel_J /*channel-last*/, int D, int H,
                           int W, const float* tfs, const int32_t* bone_ids, const float* offset, const float* scale,
                           uint64_t* counters, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* tinycudann.Encoding replacements (reference call sites models/network_utils.py:65,77,191;
 * configs/geometry/progressive_hash_grid.yaml:9-24, configs/radiance/progressive_hash_grid.yaml:5-19).
 * HashGrid: n_levels x 2 features, fp32 table [entries, 2], level-major output
 * [l0f0, l0f1, l1f0, ...]; x in [0,1]^3.  out_stride lets the caller write the 32 features
 * straight into a wider row (it must be even).  dy_dx (optional) = d out / d x, [n, L*2, 3]. */
int64_t ia_hashgrid_n_entries(int n_levels, int log2_hashmap_size, int base_resolution, float per_level_scale);
int ia_hashgrid_fwd(int64_t n, const float* x /*[n,3]*/, const float* params, int n_levels, int n_features,
                    int log2_hashmap_size, int base_resolution, float per_level_scale,
                    float* out, int out_stride, float* dy_dx /*or NULL*/, ia_stream_t stream);
/* XCD-partitioned variant for large batches (same outputs): each XCD gathers from ONE level's table, which then
 * fits its 4 MiB L2, instead of sixteen; level-major intermediate in `scratch` (ia_hashgrid_fwd_scratch_bytes) +
 * a transpose pass.  Relies on the observed block -> XCD round-robin for speed only. */
int64_t ia_hashgrid_fwd_scratch_bytes(int64_t n, int n_levels, int with_jac);
int ia_hashgrid_fwd_xcd(int64_t n, const float* x, const float* params, int n_levels, int n_features,
                        int log2_hashmap_size, int base_resolution, float per_level_scale, float* out, int out_stride,
                        float* dy_dx, void* scratch, ia_stream_t stream);
/* backward w.r.t. the table (atomic accumulate into grad_params, caller zeroes it):
 *   grad[c] += g_enc[l,:] * w_c  +  g_jac[l,:] * sum_a q[a] * d w_c / d x_a
 * The second term (g_jac, q both non-NULL) is the double-backward needed when a loss depends on
 * the analytic normal d sdf / d x (eikonal, normal-conditioned radiance). */
int ia_hashgrid_bwd(int64_t n, const float* x, int n_levels, int n_features, int log2_hashmap_size,
                    int base_resolution, float per_level_scale, const float* g_enc /*[n,stride] or NULL*/,
                    int g_enc_stride, const float* g_jac /*or NULL*/, int g_jac_stride, const float* q /*[n,3] or NULL*/,
                    float* grad_params, ia_stream_t stream);
/* same result (up to fp32 summation order) without the fabric atomics: multisplit of the (index, value) records into
 * (level, 8192-entry slice) buckets + LDS reduction per bucket in 64-bit fixed point with integer atomics (exact,
 * order-independent sums of values quantised to 2^-40 of the level's largest |value|; see csrc/hashgrid.hip).  scratch: device buffer of
 * ia_hashgrid_bwd_scratch_bytes(n, ...) bytes (~1.3 KB per point); n * n_levels * 8 must be < 2^31. */
int64_t ia_hashgrid_bwd_scratch_bytes(int64_t n, int n_levels, int log2_hashmap_size, int base_resolution,
                                      float per_level_scale);
int ia_hashgrid_bwd_binned(int64_t n, const float* x, int n_levels, int n_features, int log2_hashmap_size,
                           int base_resolution, float per_level_scale, const float* g_enc, int g_enc_stride,
                           const float* g_jac, int g_jac_stride, const float* q, float* grad_params,
                           uint32_t level_mask /* bit l = level l takes part (progressive bands); ~0u = all */,
                           void* scratch, int64_t scratch_bytes, ia_stream_t stream);
/* contractions with the stored Jacobian dy_dx [n,K,3]: mode 0: out[n,3] = sum_k v[n,k] J[n,k,:] (input gradient);
 * mode 1: out[n,K] = J[n,k,:] . v[n,:3] (JVP).  Used by the tinycudann.Encoding drop-in's (double) backward. */
int ia_hashgrid_jac_contract(int mode, int64_t n, int K, const float* jac, const float* v, int v_stride, float* out,
                             int out_stride, ia_stream_t stream);
/* SphericalHarmonics(degree=4): d01 in [0,1]^3 (tcnn maps to [-1,1]) -> 16 values */
int ia_sh4_fwd(int64_t n, const float* d01, float* out, int out_stride, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Fused MLP evaluation on the matrix cores (fp32 MFMA).  kind:
 *   0  SDF       35 -> 64 -> 13, Softplus(beta=100)   models/network_utils.py:201-244, rf/geometry.py:152-172
 *   1  radiance  67 -> 64 -> 64 -> 3, ReLU, sigmoid   models/rf/radiance.py:111-135
 *   2  material  48 -> 64 -> 64 -> 5, ReLU, sigmoid   models/pbr/material.py:31-51, network_utils.py:410-428
 * The input row is the concatenation of n_segs (<=5) sources, each `seg_width[s]` columns read
 * from seg_ptr[s] with row stride seg_stride[s] and mapped v*mul+add.  Weights are the EFFECTIVE
 * row-major matrices (weight-norm / Lipschitz scaling / level masks / column order folded in by the
 * host).  seg_* and inv_scale_host are HOST arrays; everything else is device memory.
 * kind 0 with grad != NULL also returns the analytic d sdf / d x (needs jac = hash-grid dy_dx,
 * segment 0 = the 32 hash features, xyz_col = first xyz column). */
int ia_mlp_fwd(int kind, int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
               const int* seg_width, const float* seg_mul, const float* seg_add,
               const float* W1, const float* b1, const float* W2, const float* b2, const float* Wo, const float* bo,
               float* y, int y_stride, const float* jac, int xyz_col, const float* inv_scale_host, float* grad,
               ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* SNARF candidate bookkeeping + per-sample shading prep (replaces the mask-index / gather glue of
 * models/deformers/snarf_deformer.py:187-261 and models/intrinsic_avatar.py:1032-1064,390-394).    */
/* filter (K9) fused with the per-point count of surviving candidates */
int ia_deform_filter_count(int64_t P, int I, const float* x /*[P,I,3]*/, const uint8_t* valid /*[P,I]*/,
                           uint8_t* mask /*[P,I]*/, int32_t* cnt /*[P]*/, ia_stream_t stream);
/* packed candidate list in (point, init) order; start = exclusive scan of cnt */
int ia_deform_compact(int64_t P, int I, const float* x, const uint8_t* mask, const int32_t* start,
                      float* cand_x /*[Q,3]*/, int32_t* cand_src /*[Q] = p*I+i*/, ia_stream_t stream);
/* the three steps above (filter + count, exclusive scan, packed list) in one pass over x: a tile of points is filtered in
 * LDS and written to its final place, the tile offsets come from a chained (look-back) scan.  cnt / start [P] as above,
 * *total = Q.  cand_x must hold the packed list; it MAY BE x ITSELF (in-place compaction: the packed list is a prefix of
 * x's storage, x is consumed).  cand_src and mask are optional (NULL).  tmp: ia_deform_filter_compact_tmp_bytes(P) bytes,
 * 8-byte aligned.  Deterministic: identical outputs to the three separate calls. */
size_t ia_deform_filter_compact_tmp_bytes(int64_t P);
int ia_deform_filter_compact(int64_t P, int I, const float* x, const uint8_t* valid, int32_t* cnt /*[P]*/, int32_t* start /*[P]*/,
                             float* cand_x /*[Q,3], or x*/, int32_t* cand_src /*[Q] or NULL*/, uint8_t* mask /*[P,I] or NULL*/,
                             int32_t* total /*[1]*/, void* tmp, size_t tmp_bytes, ia_stream_t stream);
/* the same packing in two steps with no dependence between tiles (faster: the look-back above runs at half the rate of its
 * loads).  ia_deform_filter_tiles: filter + count, every 256-point tile's candidates packed IN PLACE at the start of the tile's
 * own rows of x (x is consumed; (point, init) indices likewise into src_local [P*I], optional), cnt [P], tile-local start [P],
 * *total = Q.  The host reads Q and allocates cand_x [Q,3]; ia_deform_pack_tiles copies the tile blocks there (cand_x must not
 * alias x; cand_src [Q] iff src_local) and makes start global.  tmp (ia_deform_filter_tiles_tmp_bytes(P) bytes, 8-byte aligned)
 * carries the tile offsets from the first call to the second. */
size_t ia_deform_filter_tiles_tmp_bytes(int64_t P);
int ia_deform_filter_tiles(int64_t P, int I, float* x, const uint8_t* valid, int32_t* cnt, int32_t* start, int32_t* src_local /*or NULL*/,
                           uint8_t* mask /*or NULL*/, int32_t* total, void* tmp, size_t tmp_bytes, ia_stream_t stream);
int ia_deform_pack_tiles(int64_t P, int I, const float* x, const int32_t* src_local /*or NULL*/, int32_t* start, float* cand_x,
                         int32_t* cand_src /*or NULL*/, const void* tmp, ia_stream_t stream);
/* first-minimum SDF over each point's candidates (1e5 / zeros / [0,0,1] defaults when none),
 * gradient pushed to posed space with c2w[cand_src] (3x3, row-major) */
int ia_deform_select(int64_t P, const int32_t* start, const int32_t* cnt, const float* cand_x,
                     const int32_t* cand_src, const float* cand_sdf, int sdf_stride, const float* cand_grad /*or NULL*/,
                     const float* cand_feat /*or NULL*/, int feat_stride, int feat_dim, const float* c2w /*[P*I,9] or NULL*/,
                     float* pts_cano /*[P,3]*/, float* sdf /*[P]*/, uint8_t* valid /*[P]*/, int32_t* sel /*[P] or NULL*/,
                     float* grad_posed /*[P,3] or NULL*/, float* grad_cano /*[P,3] or NULL*/, float* feat /*[P,feat_dim] or NULL*/,
                     ia_stream_t stream);
/* pts = o[ray] + d[ray] * t,  t = t0 (t1 NULL) or (t0+t1)/2 */
int ia_ray_points(int64_t n, const float* rays_o, const float* rays_d, const int64_t* ray_indices,
                  const float* t0, const float* t1, float* pts, ia_stream_t stream);
/* normals (SMPL + world), reflected view direction mapped to [0,1] for the SH encoding */
int ia_shade_prep(int64_t n, const float* sdf_grad /*[n,3]*/, const float* rays_d, const int64_t* ray_indices,
                  const float* w2s_rot /*[9] device*/, float* normal_smpl, float* normal_world, float* refl01,
                  ia_stream_t stream);
/* alpha = 1 - exp(-LaplaceDensity(sdf; beta) * dist); dists NULL => dist_const; beta: 1 device float */
int ia_laplace_alpha(int64_t n, const float* sdf, const float* dists, float dist_const, const float* beta,
                     float* alpha, ia_stream_t stream);
/* ... with the interval length taken from the interval's ends: dist = t_ends[i] - t_starts[i] (same fp32 subtraction, not materialised) */
int ia_laplace_alpha_intervals(int64_t n, const float* sdf, const float* t_starts, const float* t_ends, const float* beta, float* alpha,
                               ia_stream_t stream);
int ia_laplace_alpha_bwd(int64_t n, const float* sdf, const float* dists, float dist_const, const float* beta,
                         const float* g_alpha, float* g_sdf, float* g_beta /*1 float, accumulated*/, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Backward (training step).  The reference obtains these from torch autograd through VanillaMLP /
 * LipshitzMLP / tcnn, including the double backward through the analytic normal
 * (models/rf/geometry.py:165-172, create_graph=True). */
int ia_sh4_bwd(int64_t n, const float* d01, const float* g_sh, int g_stride, float* g_d01, ia_stream_t stream);
int ia_shade_prep_bwd(int64_t n, const float* sdf_grad, const float* rays_d, const int64_t* ray_indices,
                      const float* w2s_rot, const float* g_normal_world /*or NULL*/, const float* g_refl01 /*or NULL*/,
                      const float* g_normal_smpl /*or NULL: gradient of g / max(|g|, 1e-6)*/, float* g_sdf_grad, ia_stream_t stream);
/* data-path backward of the two-hidden-layer ReLU MLPs (kind 1 radiance, 2 material; sigmoid output):
 * g_x [n, gx_stride] = d L / d assembled input row; X/A1/A2 (layer inputs) and G1/G2/G3 (pre-activation
 * gradients) are emitted for the weight-gradient GEMMs dW_l = G_l^T A_{l-1} (plain library GEMM). */
int ia_mlp_bwd(int kind, int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
               const int* seg_width, const float* seg_mul, const float* seg_add,
               const float* W1, const float* b1, const float* W2, const float* b2, const float* Wo, const float* bo,
               const float* g_y /*[n,OUT]*/, float* g_x, int gx_stride,
               float* X /*[n,IN_PAD]*/, float* A1 /*[n,64]*/, float* A2 /*[n,64]*/,
               float* G1 /*[n,64]*/, float* G2 /*[n,64]*/, float* G3 /*[n,16]*/, ia_stream_t stream);
/* SDF head with first- and second-order terms (see csrc/mlp_bwd.hip):
 * in : g_out [n,13] = d L / d out, q [n,3] = d L / d (d sdf / d x') ; jac = hash-grid dy_dx
 * out: gE, gG [n,32] for ia_hashgrid_bwd(g_enc = gE, g_jac = gG, q); operand pairs
 *      dW1 = DZ^T Hh + GZ^T U, db1 = sum DZ, dW2 = g_out^T A (+ row 0 += sum DGS), db2 = sum g_out */
int ia_sdf_mlp_bwd(int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride, const int* seg_width,
                   const float* seg_mul, const float* seg_add, const float* W1, const float* b1, const float* Wo,
                   const float* bo, const float* jac, const float* g_out, const float* q, float* gE, float* gG,
                   float* Hh /*[n,36]*/, float* U /*[n,36]*/, float* DZ /*[n,64]*/, float* GZ /*[n,64]*/,
                   float* A /*[n,64]*/, float* DGS /*[n,64]*/, ia_stream_t stream);

/* Fused training backward (csrc/mlp_train.hip): the same gradients as ia_mlp_bwd / ia_sdf_mlp_bwd + ia_wgrad, but the
 * layer operands never leave LDS and the weight gradients are accumulated in MFMA accumulator registers across the whole
 * batch.  All d* buffers are ACCUMULATED INTO (caller zeroes): dW1 [64,IN], db1 [64], dW2 [64,64], db2 [64],
 * dWo [OUT,64], dbo [OUT] (row-major, effective-weight layout).  g_x [n, gx_stride] receives d L / d input row. */
int ia_mlp_bwd_fused(int kind, int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
                     const int* seg_width, const float* seg_mul, const float* seg_add, const float* W1, const float* b1,
                     const float* W2, const float* b2, const float* Wo, const float* bo, const float* g_y, float* g_x,
                     int gx_stride, float* dW1, float* db1, float* dW2, float* db2, float* dWo, float* dbo,
                     ia_stream_t stream);
int ia_sdf_mlp_bwd_fused(int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride, const int* seg_width,
                         const float* seg_mul, const float* seg_add, const float* W1, const float* b1, const float* Wo,
                         const float* bo, const float* jac, const float* g_out, const float* q, float* gE, float* gG,
                         float* dW1 /*[64,35]*/, float* db1, float* dWo /*[13,64]*/, float* dbo,
                         float* g_xyz /*[n,3] or NULL: first-order gradient w.r.t. the xyz input columns (pose gradients)*/,
                         ia_stream_t stream);

/* split-K weight gradient on the matrix cores: dW[M,ldw] += G[:, :M]^T . A[:, :N], db[M] += column sums of G
 * (M <= 64, N <= 96; strides <= 64 / 96 floats; accumulates with atomics into caller-zeroed dW / db; db may be NULL) */
int ia_wgrad(int64_t n, const float* G, int g_stride, int M, const float* A, int a_stride, int N, float* dW, int ldw,
             float* db, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* lib.torch_pbr call chain of pbr_light_forward (models/intrinsic_avatar.py:755-861), fused:
 * scatterer.eval (Lambert + GGX multi-lobe, incl. cosine) + emitter.eval + emitter.pdf + Li/Lo assembly for F
 * foreground shading samples.  normals / view_dirs / light_dirs are SMPL-space; w2s_rot = w2s[:3,:3] (device, 9
 * floats) maps light dirs to world for the equirect lookup; transmittance [F] and indirect_rgb [F,3] (or NULL) come
 * from compute_indirect_radiance.  env_base [H,W,3], env_pmf [H,W] (luminance x sin(theta), sums to 1). */
int ia_pbr_light_shade(int64_t F, const float* normal, const float* albedo, const float* roughness,
                       const float* metallic, const float* view_dirs, const float* light_dirs,
                       const float* transmittance, const float* indirect_rgb, const float* env_base,
                       const float* env_pmf, int env_h, int env_w, const float* w2s_rot,
                       float* Lo, float* Lo_diff, float* Lo_spec, ia_stream_t stream);
/* all four estimators of the reference with one kernel: mode 0 light (:755-861, == ia_pbr_light_shade), 1 uniform_light
 * (:654-753; inv_pdf [F] from the stratified sphere, vis [F,3] = 2*tr output), 2 mis (:547-652; weight 1/(pdf_scatter +
 * pdf_light), call once on the 2F concatenated scatter+light directions and sum the halves), 3 mats (:863-948). */
int ia_pbr_shade(int mode, int64_t F, const float* normal, const float* albedo, const float* roughness,
                 const float* metallic, const float* view_dirs, const float* out_dirs, const float* transmittance,
                 const float* indirect_rgb, const float* inv_pdf, const float* env_base, const float* env_pmf,
                 int env_h, int env_w, const float* w2s_rot, float* Lo, float* Lo_diff, float* Lo_spec, float* vis,
                 ia_stream_t stream);
/* backward of modes 0 / 1 w.r.t. the per-point surface attributes and the environment texels (autograd through
 * scatterer.eval + emitter.eval in the reference; directions, transmittance, masks and sampling weights are constants).
 * g_Lo / g_Lo_diff / g_Lo_spec: incoming gradients of the three outputs (any may be NULL); g_env_base [H,W,3] is
 * accumulated into with atomics (may be NULL). */
int ia_pbr_shade_bwd(int mode, int64_t F, const float* normal, const float* albedo, const float* roughness,
                     const float* metallic, const float* view_dirs, const float* out_dirs, const float* transmittance,
                     const float* indirect_rgb, const float* inv_pdf, const float* env_base, const float* env_pmf,
                     int env_h, int env_w, const float* w2s_rot, const float* g_Lo, const float* g_Lo_diff,
                     const float* g_Lo_spec, float* g_normal, float* g_albedo, float* g_roughness, float* g_metallic,
                     float* g_env_base, void* scratch /*or NULL*/, size_t scratch_bytes, ia_stream_t stream);
/* bytes of scratch (16-byte aligned) with which ia_pbr_shade_bwd accumulates g_env_base through per-sample records and LDS
 * instead of 12 device-scope atomics per sample; 0 = the batch is too small to benefit (pass NULL). */
size_t ia_pbr_shade_bwd_scratch_bytes(int64_t F);
/* scatterer.sample / scatterer.pdf of the multi-lobe BRDF (1/2 cosine hemisphere + 1/2 GGX half-vector sampling);
 * u [F,3] uniforms = (lobe selector, u1, u2): explicit RNG */
int ia_brdf_sample(int64_t F, const float* normal, const float* view_dirs, const float* roughness, const float* u,
                   float* out_dirs, ia_stream_t stream);
int ia_brdf_pdf(int64_t F, const float* normal, const float* view_dirs, const float* out_dirs, const float* roughness,
                float* pdf, ia_stream_t stream);
/* emitter.eval / emitter.pdf on world-space unit directions (either output may be NULL) */
int ia_envlight_eval(int64_t n, const float* dirs_world, const float* env_base, const float* env_pmf, int env_h,
                     int env_w, float* rgb /*[n,3]*/, float* pdf /*[n]*/, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Occupancy-grid maintenance (TemporalOccGridEstimator._update, models/occ_grid/temporal_occ_grid.py:369-411;
 * _compute_occupancy_grid, models/intrinsic_avatar.py:307-358; max_connected_component, models/utils.py:152-163). */
/* occs = max(occs * ema_decay, occ_new) */
int ia_occgrid_ema(int64_t n_cells, float* occs, const float* occ_new, float ema_decay, ia_stream_t stream);
int64_t ia_occgrid_tmp_bytes(int res_x, int res_y, int res_z);
/* 3^3 max-pool dilation -> thre = min(mean(pooled >= 0), thre_max) -> binaries = pooled > thre ->
 * (optional) keep the largest 26-connected component (res_z*3 label-propagation sweeps, most frequent label).
 * thre_out: 1 device float or NULL. */
int ia_occgrid_binarize(int res_x, int res_y, int res_z, const float* occs, float thre_max, int keep_largest_component,
                        uint8_t* binaries, float* thre_out, void* tmp, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Elementwise tail of SNARFDeformer.deform for the winning candidates (snarf_deformer.py:192-231) and the eikonal term
 * (systems/intrinsic_avatar.py:195-203), one kernel per direction instead of ~20 torch launches over the samples.
 *   c2w[i]      = fwd_J[cand_src[max(sel[i], 0)]]          (zeros when fwd_J == NULL)
 *   feat[i]     = valid ? out13[i] : 0;   sdf[i] = valid ? out13[i,0] : 1e5;
 *   sdf_grad[i] = valid ? c2w[i] . grad_c[i] : (0, 0, 1)
 * _bwd: g_out13 = valid ? g_feat (+ g_sdf on column 0) : 0;  g_grad_c = valid ? c2w^T g_sdf_grad : 0  (NULL gradients = 0). */
int ia_select_push(int64_t n, const float* out13, const float* grad_c, const uint8_t* valid, const float* fwd_J,
                   const int32_t* cand_src, const int32_t* sel, float* feat, float* sdf, float* sdf_grad, float* c2w,
                   ia_stream_t stream);
int ia_select_push_bwd(int64_t n, const uint8_t* valid, const float* c2w, const float* g_feat, const float* g_sdf,
                       const float* g_sdf_grad, float* g_out13, float* g_grad_c, ia_stream_t stream);
/* partial[k] = (sum over the valid samples of workgroup k of (|sdf_grad| - 1)^2, number of valid samples);
 * ia_eikonal_partials(n) pairs.  _bwd: g = weight[0] * 2 (|g| - 1) g / |g| on valid samples, 0 elsewhere. */
int64_t ia_eikonal_partials(int64_t n);
int ia_eikonal(int64_t n, const float* sdf_grad, const uint8_t* valid, float* partial, ia_stream_t stream);
int ia_eikonal_bwd(int64_t n, const float* sdf_grad, const uint8_t* valid, const float* weight, float* g_sdf_grad,
                   ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Optimiser step.  torch.optim.Adam exactly as the reference builds it (configs/config.yaml:110-136,
 * systems/utils.py:314-325: Adam, betas (0.9, 0.99), eps 1e-15, per-group lr and L2 weight_decay), for n_tensors
 * parameter tensors in ONE launch (40 per launch internally).  params/grads/exp_avg/exp_avg_sq/numel/step_size/
 * weight_decay are HOST arrays of length n_tensors; the pointers they hold are device memory (fp32, contiguous).
 *   g = grad * grad_scale (+ weight_decay p);  m += (g - m)(1 - beta1);  v = v beta2 + (1 - beta2) g g;
 *   p -= step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * step_size[t] = lr_t / (1 - beta1^step) and bias_correction2_sqrt = sqrt(1 - beta2^step) are formed by the caller in
 * double precision, as torch/optim/adam.py:_single_tensor_adam does.  grad_scale folds in DDP's 1/world averaging. */
int ia_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const int64_t* numel, const float* step_size, const float* weight_decay,
                 float beta1, float beta2, float eps, float bias_correction2_sqrt, float grad_scale, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Host logic between K1 and the PBR estimator (models/pbr/utils.py:130-229 sample_volume_interaction;
 * models/intrinsic_avatar.py:1356-1378 light shuffle, :1335-1342,:1420-1466 Lo scatter + accumulate, :788-803 secondary-ray
 * mask; lib.torch_pbr emitter.sample).  K1's output is structured: a ray with samples owns spp consecutive re-samples, its
 * foreground ones first (j < spp - bg_counts[ray]) in non-decreasing order of the sampled interval.  The foreground list
 * [F] is ray-major; fg_start = exclusive scan of fg_ray_cnt; the re-samples of interval s are the range
 * [fg_off[s], fg_off[s] + fg_counts[s]) of it (fg_off = exclusive scan of fg_counts). */
int ia_vi_layout(int64_t n_rays, int spp, const int32_t* resampled_packed_info, const int32_t* bg_counts, int32_t* fg_ray_cnt,
                 ia_stream_t stream);
/* gathers of the per-interval attributes + positions o + d t, view dirs, re-sampled fg weights w[s] / fg_counts[s] */
int ia_vi_gather(int64_t n_rays, const int32_t* resampled_packed_info, const int32_t* fg_ray_cnt, const int32_t* fg_start,
                 const float* ts /*[R]*/, const int64_t* sampled_idx /*[R]*/, const int32_t* fg_counts /*[S]*/,
                 const float* weights /*[S]*/, const float* rays_o, const float* rays_d, const float* normals /*[S,3]*/,
                 const float* albedo /*[S,3]*/, const float* roughness /*[S]*/, const float* metallic /*[S]*/,
                 int32_t* fg_src /*[F]*/, int32_t* fg_ray /*[F]*/, float* positions /*[F,3]*/, float* view_dirs /*[F,3]*/,
                 float* o_normals, float* o_albedo, float* o_roughness, float* o_metallic, float* o_weights /*[F]*/,
                 ia_stream_t stream);
/* backward of the gathers: segmented sums over each interval's contiguous fg range (no atomics, deterministic) */
int ia_vi_gather_bwd(int64_t S, const int32_t* fg_counts, const int32_t* fg_off, const float* g_normals_fg,
                     const float* g_albedo_fg, const float* g_roughness_fg, const float* g_metallic_fg,
                     const float* g_weights_fg, float* g_normals, float* g_albedo, float* g_roughness, float* g_metallic,
                     float* g_weights, ia_stream_t stream);
/* rgb[r] = sum_fg w Lo + [bg_counts > 0] transmittance[r] * background (rays without samples: background);
 * background_rays [n,3] (optional) overrides the constant colour per ray (add_emitter) */
int ia_vi_composite(int64_t n_rays, const int32_t* resampled_packed_info, const int32_t* fg_ray_cnt, const int32_t* fg_start,
                    const int32_t* bg_counts, const float* weights_fg, const float* Lo /*[F,3]*/, const float* transmittance /*[n]*/,
                    const float* background /*[3]*/, const float* background_rays, float* rgb /*[n,3]*/, ia_stream_t stream);
int ia_vi_composite_bwd(int64_t n_rays, int64_t F, const int32_t* resampled_packed_info, const int32_t* bg_counts,
                        const int32_t* fg_ray, const float* weights_fg, const float* Lo, const float* background,
                        const float* background_rays /* as passed to the forward */, const float* g_rgb, float* g_weights_fg,
                        float* g_Lo, float* g_transmittance, ia_stream_t stream);
/* the reference's index lists: fg_indices [F], bg_indices [R-F], resampled_ray_indices [R], resampled_weights [R] (each optional) */
int ia_vi_indices(int64_t n_rays, int spp, const int32_t* resampled_packed_info, const int32_t* fg_ray_cnt,
                  const int32_t* fg_start, const int32_t* bg_counts, const int64_t* sampled_idx, const int32_t* fg_counts,
                  const float* weights, const float* transmittance, int64_t* fg_indices, int64_t* bg_indices,
                  int64_t* ray_indices, float* resampled_weights, ia_stream_t stream);
/* per-ray permutation of [0, spp) = stable argsort of u[r, :] (explicit uniforms in [0,1)), first fg_ray_cnt[r] entries
 * written at fg_start[r] (models/intrinsic_avatar.py:1356-1378) */
int ia_light_shuffle(int64_t n_rays, int spp, const int32_t* fg_ray_cnt, const int32_t* fg_start, const float* u /*[n,spp]*/,
                     int32_t* shuffled /*[F]*/, ia_stream_t stream);
/* emitter.sample: k directions ~ luminance x sin(theta): inverse CDF of the flattened pmf (u[:,0]), jitter in the texel
 * (u[:,1:3]); rot [9] (optional) = world -> SMPL rotation applied + normalised (transform_dirs_w2s) */
int ia_envlight_sample(int64_t k, const float* u /*[k,3]*/, const double* cdf /*[H*W]*/, int env_h, int env_w, const float* rot,
                       float* dirs /*[k,3]*/, ia_stream_t stream);
/* secondary rays of the light estimators (:788-803): flag = n . d > 1e-6 (d = dirs[dir_index[k]] when dir_index is given),
 * compaction into ray lists (slot = exclusive scan of flag), scatter of the traced results back (transmittance clamped) */
int ia_secondary_mask(int64_t F, const float* normals, const float* dirs, const int32_t* dir_index, int32_t* flag,
                      ia_stream_t stream);
int ia_secondary_compact(int64_t F, const int32_t* flag, const int32_t* slot, const float* positions, const float* dirs,
                         const int32_t* dir_index, float* rays_o, float* rays_d, int32_t* src, float* dense_dirs /*[F,3] or NULL*/,
                         ia_stream_t stream);
int ia_secondary_scatter(int64_t M, const int32_t* src, const float* transmittance, const float* rgb, float* dense_transmittance,
                         float* dense_rgb, ia_stream_t stream);
/* ... the same result written point by point from (flag, slot): every dense element is written exactly once, no zero fill beforehand */
int ia_secondary_gather_dense(int64_t F, const int32_t* flag, const int32_t* slot, const float* transmittance, const float* rgb,
                              float* dense_transmittance, float* dense_rgb, ia_stream_t stream);

/* lib.torch_pbr scatterer classes (registered at models/__init__.py:44-50; call sites models/intrinsic_avatar.py:566-574,
 * 591-614, 714-723, 816-825, 882-923): sample / pdf / eval of the lobe set `lobes` = 1 Lambertian, 2 GGX, 3 MultiLobe,
 * 4 Mirror.  wi points away from the surface; eval returns (diff [F], spec [F,3]) including the cosine term. */
int ia_scatterer_sample(int64_t F, int lobes, const float* normal, const float* wi, const float* alpha, const float* u /*[F,3]*/,
                        float* wo, ia_stream_t stream);
int ia_scatterer_pdf(int64_t F, int lobes, const float* normal, const float* wi, const float* wo, const float* alpha, float* pdf,
                     ia_stream_t stream);
int ia_scatterer_eval(int64_t F, int lobes, const float* normal, const float* wi, const float* wo, const float* alpha,
                      const float* albedo, const float* metallic, float* diff, float* spec, ia_stream_t stream);

/* spatial ordering of large query batches (no reference counterpart: a scheduling aid, results are order-independent):
 * 30-bit Morton code of each point's cell for a key-value sort, and the gather / scatter that apply the permutation */
int ia_morton_keys(int64_t n, const float* pts /*[n,3]*/, const float* origin_host3 /*HOST pointer to 3 floats*/, float inv_cell,
                   int32_t* keys, ia_stream_t stream);
int ia_gather_rows3(int64_t n, const float* src, const int64_t* order, float* dst, ia_stream_t stream);
int ia_scatter_f32(int64_t n, const float* src, const int64_t* order, float* dst, ia_stream_t stream);
/* the same ordering in one call: order [n] int32 = stable argsort of the Morton codes restricted to their bits [drop_bits, 30)
 * (hand-written LSD radix sort of (key, index) pairs, csrc/sort.hip: three 10-bit passes of count -> scan -> ranked scatter, no
 * workgroup waits for another one), with the int32 forms of the gather / scatter.  tmp: ia_morton_order_tmp_bytes(n) bytes,
 * 256-byte aligned.  ia_sort_rank_mode(): 1 = in-wave ranks by LDS atomics with return (checked on the device at the first sort),
 * 2 = by ballots (IA_SORT_RANK=ballot or the check failed), 0 = no sort has run yet. */
int ia_sort_rank_mode(void);
size_t ia_morton_order_tmp_bytes(int64_t n);
int ia_morton_order(int64_t n, const float* pts /*[n,3]*/, const float* origin_host3, float inv_cell, int drop_bits,
                    int32_t* order /*[n]*/, void* tmp, size_t tmp_bytes, ia_stream_t stream);
int ia_gather_rows3_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream);
int ia_scatter_f32_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream);
int ia_scatter_rows3_i32(int64_t n, const float* src /*[n,3]*/, const int32_t* order, float* dst /*[n,3]: dst[order[i]] = src[i]*/,
                         ia_stream_t stream);

/* GaussianHistogram (models/utils.py:133-149) of the albedo-entropy regulariser (models/pbr/material.py:59-70):
 * out[b] (caller-zeroed, accumulated) = sum_n exp(-0.5 ((x_n - c_b) / sigma)^2) / (sigma sqrt(2 pi)) * delta; sigma is a
 * DEVICE scalar (the reference passes torch.var(channel)); backward w.r.t. x [n] and sigma (caller-zeroed scalar). */
int ia_gaussian_histogram(int64_t n, const float* x, const float* sigma, int bins, float vmin, float vmax, float* out,
                          ia_stream_t stream);
int ia_gaussian_histogram_bwd(int64_t n, const float* x, const float* sigma, int bins, float vmin, float vmax, const float* g_out,
                              float* g_x, float* g_sigma, ia_stream_t stream);

/* SDF-only query path of the no-grad coarse passes (coarse_alpha_fn / alpha_fn / coarse_alpha_sdf_fn, models/intrinsic_avatar.py
 * :399-428, :955-1030): ia_hashgrid_fwd_xcd with out == NULL leaves its level-major result (float2 [L][n]) in `scratch`;
 * ia_sdf_levels_fwd runs the SDF head on it and writes the SDF alone; ia_deform_select_min = min over a point's candidates
 * (1e5 when it has none, snarf_deformer.py:192). */
int ia_sdf_levels_fwd(int64_t n, const void* levels, const float* xp /*[n,3] in [0,1]*/, const float* W1 /*[64,35] hash|xyz*/,
                      const float* b1, const float* Wo /*[>=1,64], row 0 used*/, const float* bo, float* sdf /*[n]*/,
                      ia_stream_t stream);
/* the general SDF head on level-major gather results (no [n,32] rows, no [n,32,3] Jacobian): ia_hashgrid_fwd_levels = the
 * XCD-partitioned gather without its transpose (scratch: ia_hashgrid_fwd_scratch_bytes(n, L, with_jac) bytes; features float2
 * [L][n] at its start, Jacobian float [L][n][6] at ia_hashgrid_fwd_levels_jac_offset(n, L)); ia_sdf_levels_fwd_grad = kind 0 of
 * ia_mlp_fwd with the analytic gradient (y [n,13], grad [n,3]) reading both. */
int64_t ia_hashgrid_fwd_levels_jac_offset(int64_t n, int n_levels);
int ia_hashgrid_fwd_levels(int64_t n, const float* x, const float* params, int n_levels, int n_features, int log2_hashmap_size,
                           int base_resolution, float per_level_scale, int with_jac, void* scratch, ia_stream_t stream);
int ia_sdf_levels_fwd_grad(int64_t n, const void* levels, const float* levels_jac, const float* xp, const float* W1, const float* b1,
                           const float* Wo /*[13,64]*/, const float* bo, float* y, int y_stride, const float* inv_scale_host,
                           float* grad, ia_stream_t stream);
int ia_deform_select_min(int64_t P, const int32_t* start, const int32_t* cnt, const float* cand_sdf, float* sdf, ia_stream_t stream);
/* ... for points evaluated as a permutation of the caller's list: sdf[order[p]] = the minimum of point p */
int ia_deform_select_min_scatter(int64_t P, const int32_t* start, const int32_t* cnt, const float* cand_sdf, const int32_t* order,
                                 float* sdf, ia_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Per-step operators of a training step (csrc/stepops.hip; SURVEY 8(f) row 2: host orchestration).  Each replaces a chain of
 * element-wise / reduce operators that the reference issues through torch (and their autograd backward chains) by one launch each way;
 * a 4096-ray training batch (configs/sampler/edge.yaml:2) is bound by the host's launch rate, not by a kernel.
 *
 * ia_normalize_points: out = (x - center) / scale + 0.5 -- the [0,1]^3 coordinates of a hash grid (models/rf/geometry.py:155,
 *   models/rf/radiance.py:115).  x, out [n,3]; center, scale [3] on the device.
 * ia_effective_weights(_bwd): the weight matrix a fused MLP kernel reads, from the parameters of one linear layer.
 *   mode 0 plain Linear (models/network_utils.py:232-244), 1 weight norm  g * v / |v|_row (nn.utils.weight_norm, :201-244; g [M]),
 *   2 Lipschitz normalisation v * min(softplus(c) / sum_row |v|, 1) (LipshitzMLP, :396-403; g = c, ONE scalar).
 *   src [N] int32 (NULL: identity): parameter column of output column j (the kernels' column order [hash | xyz | ...]);
 *   mul [N] (NULL: 1): mask of output column j (ProgressiveBandHashGrid level mask :79-100, SH band mask models/rf/radiance.py:140-155).
 *   _bwd: g_out [M,N] -> g_v [M,N] (parameter column order; every column must appear in src) and g_g ([M] mode 1, [1] mode 2).
 * ia_sg_image(_bwd): EnvironmentLightSG.generate_image (lib/torch_pbr, absent; call site models/intrinsic_avatar.py:281-305):
 *   out[row, col] = sum_k softplus(mu_k) exp(exp(log_lambda_k) (d(row, col) . normalize(axis_k) - 1)), d = the equirectangular direction
 *   convention of ia_envlight_eval; out [H,W,3].  _bwd: g_img -> g_axis [K,3], g_log_lambda [K], g_mu [K,3]; tmp: ia_sg_image_bwd_tmp_bytes(K).
 * ia_envlight_pdf_tables: emitter.update_pdf (:777-781): pmf [H*W] fp32 = luminance x sin(theta), normalised in double; cdf [H*W] double =
 *   running sum of the fp32 pmf (what ia_envlight_sample searches); tmp: ia_envlight_pdf_tables_tmp_bytes(H, W) bytes; three launches.
 * ia_uniform_sphere_stratified: emitter.sample_uniform_sphere_stratified(n_rays, 16, 32) (:680-689): u [n_theta*n_phi,2] ->
 *   dirs [n_theta*n_phi,3] (z = 1 - 2 (i + u0) / n_theta, phi = 2 pi (j + u1) / n_phi), inv_pdf [n_theta*n_phi] = 4 pi.
 * ia_material_affine(_bwd): VolumeMaterial.forward's output ranges (models/pbr/material.py:44-50): m [n,5] sigmoid outputs ->
 *   albedo [n,3], roughness [n], metallic [n]; _bwd: NULL gradient = zero.
 * ia_phys_loss(_bwd): the default loss composition of IntrinsicAvatarSystem.training_step (systems/intrinsic_avatar.py:165-252):
 *   terms[0] = mean |comp_rgb - target|, [1] = mean |comp_rgb_phys - target|, [2] = BCE(clamp(opacity, 1e-3, 1 - 1e-3), mask),
 *   [3] = sum of eik_partials[k][0] (ia_eikonal), [4] = t0 + lambda_eik t3 / eik_denom + lambda_mask t2 + lambda_phys t1.  NULL
 *   comp_rgb_phys / target_mask / eik_partials drop their term.  _bwd: g_loss [1] (device) -> gradients of the three maps and g_eik_sum [1].
 * ia_edge_min_sdf: coarse_alpha_fn's interval SDF (models/intrinsic_avatar.py:980-990): out[i] = is_left[i] ? min(sdf[i], sdf[i+1]) : 1e10. */
int ia_normalize_points(int64_t n, const float* x, const float* center, const float* scale, float* out, ia_stream_t stream);
int ia_effective_weights(int mode, int M, int N, const float* g, const float* v, const int* src, const float* mul, float* out,
                         ia_stream_t stream);
int ia_effective_weights_bwd(int mode, int M, int N, const float* g, const float* v, const int* src, const float* mul, const float* g_out,
                             float* g_v, float* g_g, ia_stream_t stream);
int ia_sg_image(int K, int H, int W, const float* axis, const float* log_lambda, const float* mu, float* out, ia_stream_t stream);
int64_t ia_sg_image_bwd_tmp_bytes(int K);
int ia_sg_image_bwd(int K, int H, int W, const float* axis, const float* log_lambda, const float* mu, const float* g_img, void* tmp,
                    float* g_axis, float* g_log_lambda, float* g_mu, ia_stream_t stream);
int64_t ia_envlight_pdf_tables_tmp_bytes(int H, int W);
int ia_envlight_pdf_tables(int H, int W, const float* base, float* pmf, double* cdf, void* tmp, ia_stream_t stream);
int ia_uniform_sphere_stratified(int n_theta, int n_phi, const float* u, float* dirs, float* inv_pdf, ia_stream_t stream);
int ia_material_affine(int64_t n, const float* m, float albedo_scale, float albedo_bias, float roughness_scale, float roughness_bias,
                       float metallic_scale, float metallic_bias, float* albedo, float* roughness, float* metallic, ia_stream_t stream);
int ia_material_affine_bwd(int64_t n, const float* g_albedo, const float* g_roughness, const float* g_metallic, float albedo_scale,
                           float roughness_scale, float metallic_scale, float* g_m, ia_stream_t stream);
int64_t ia_phys_loss_tmp_bytes(int64_t n);
int ia_phys_loss(int64_t n, const float* comp_rgb, const float* comp_rgb_phys, const float* opacity, const float* target_rgb,
                 const float* target_mask, const float* eik_partials, int eik_k, float lambda_phys, float lambda_mask, float lambda_eik,
                 float eik_denom, float* terms, void* tmp /*ia_phys_loss_tmp_bytes(n) bytes; may be NULL when that is 0*/, ia_stream_t stream);
int ia_phys_loss_bwd(int64_t n, const float* comp_rgb, const float* comp_rgb_phys, const float* opacity, const float* target_rgb,
                     const float* target_mask, const float* g_loss, float lambda_phys, float lambda_mask, float lambda_eik, float eik_denom,
                     float* g_comp_rgb, float* g_comp_rgb_phys, float* g_opacity, float* g_eik_sum, ia_stream_t stream);
int ia_edge_min_sdf(int64_t n_edges, const float* sdf, const uint8_t* is_left, float* out, ia_stream_t stream);
/* SNARFDeformer.transform_rays_w2s (models/deformers/snarf_deformer.py:128-147): rays [n, ray_stride >= 6] (o, d, ...) -> out [n,8] =
 * (o R^T + t, d R^T, |o'| - 1, |o'| + 1), w2s [4,4] row-major on the device.  variant: summation form of the 3-term products (0: fma chain
 * k = 0,1,2; 1: separate products left to right; 2: fma chain k = 2,1,0). */
int ia_transform_rays_w2s(int64_t n, const float* rays, int ray_stride, const float* w2s, int variant, float* out, ia_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IA_AMD_H */
