/*
 * ia_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C, single-threaded restatement of the algorithms on IntrinsicAvatar's
 * volumetric render_step hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped package
 * (intrinsicavatar_amd/) never imports it and has no CPU fallback.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, IEEE float, no FMA,
 * so that every float comparison that decides an integer output is
 * reproducible bit-for-bit by the HIP kernels, which are built with the same
 * contraction setting and the same expression order).
 *
 * Parity pinning (details in DESIGN.md section 3):
 *   - K1..K7  (lib/nerfacc/cuda/csrc/{cdf,pack}.cu) and K8..K10
 *     (models/deformers/fast_snarf/cuda/...): PINNED -- checked bit-exactly
 *     against golden vectors produced in the build container by executing the
 *     reference's own kernel bodies serially on the host
 *     (tests/golden/make_golden.py).
 *   - traverse_grids / render_weight_from_alpha / accumulate_along_rays
 *     (pip nerfacc==0.5.3, absent from /root/reference): PARITY UNPINNED --
 *     restated from the published algorithm; this file *defines* the exact
 *     rounding / tie rules the HIP kernels are held to.
 *   - hash grid / spherical harmonics (tiny-cuda-nn, absent): PARITY UNPINNED,
 *     restated from the published Instant-NGP definitions.
 *
 * Every function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* T1: nerfacc.traverse_grids (call sites: models/occ_grid/temporal_occ_grid.py:166-175,
 *     models/intrinsic_avatar.py:84-93).  Upstream source: nerfacc 0.5.3
 *     nerfacc/grid.py + cuda/csrc/grid.cu + include/utils_grid.cuh (not in tree).
 * ------------------------------------------------------------------------ */

/* ray / AABB slab test; restates nerfacc utils_grid.cuh ray_aabb_intersect. */
static int ray_aabb_intersect(const float o[3], const float d[3], const float *aabb,
                              float *tmin_out, float *tmax_out)
{
    float tmin, tmax, tmin_t, tmax_t;
    if (d[0] >= 0) { tmin = (aabb[0] - o[0]) / d[0]; tmax = (aabb[3] - o[0]) / d[0]; }
    else           { tmin = (aabb[3] - o[0]) / d[0]; tmax = (aabb[0] - o[0]) / d[0]; }
    if (d[1] >= 0) { tmin_t = (aabb[1] - o[1]) / d[1]; tmax_t = (aabb[4] - o[1]) / d[1]; }
    else           { tmin_t = (aabb[4] - o[1]) / d[1]; tmax_t = (aabb[1] - o[1]) / d[1]; }
    if (tmin > tmax_t || tmin_t > tmax) return 0;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (d[2] >= 0) { tmin_t = (aabb[2] - o[2]) / d[2]; tmax_t = (aabb[5] - o[2]) / d[2]; }
    else           { tmin_t = (aabb[5] - o[2]) / d[2]; tmax_t = (aabb[2] - o[2]) / d[2]; }
    if (tmin > tmax_t || tmin_t > tmax) return 0;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (tmax <= 0) return 0;
    *tmin_out = tmin; *tmax_out = tmax;
    return 1;
}

static inline float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    float v = t * cone_angle;
    return v < dt_min ? dt_min : (v > dt_max ? dt_max : v);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One ray through ONE grid level (the reference always passes a 1-level slice:
 * temporal_occ_grid.py:169-170 `binaries[t_idx:t_idx+1]`).  first_pass counts,
 * second pass writes -- same two-pass shape as upstream. */
static void traverse_one_ray(
    int64_t tid, const float *rays_o, const float *rays_d,
    const int res[3], const uint8_t *binaries, const float *aabb,
    float near_plane, float far_plane, float step_size, float cone_angle,
    int first_pass, int64_t iv_base, int64_t sm_base,
    int64_t *iv_cnt, int64_t *sm_cnt,
    float *iv_vals, uint8_t *iv_is_left, uint8_t *iv_is_right, int64_t *iv_ray,
    float *sm_vals, int64_t *sm_ray, float *term_plane)
{
    const float *o = rays_o + 3 * tid, *d = rays_d + 3 * tid;
    int64_t n_samples = 0, n_intervals = 0;
    int continuous = 0;
    float t_last = near_plane;
    float tmin, tmax;
    const float eps = 1e-6f;

    if (ray_aabb_intersect(o, d, aabb, &tmin, &tmax)) {
        float this_tmin = fmaxf(tmin, near_plane);
        float this_tmax = fminf(tmax, far_plane);
        if (this_tmin < this_tmax) {
            /* march until t_mid is right after this_tmin */
            if (step_size <= 0.0f) t_last = this_tmin;
            else for (;;) {
                float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                if (t_last + dt * 0.5f >= this_tmin) break;
                t_last += dt;
            }
            /* setup_traversal (Amanatides & Woo) */
            float vs[3], rs[3], re[3], tdist[3], delta[3];
            int cur[3], fin[3], stp[3], ovf[3];
            for (int a = 0; a < 3; a++) {
                vs[a] = (aabb[3 + a] - aabb[a]) / (float)res[a];
                rs[a] = o[a] + d[a] * (this_tmin + eps);
                re[a] = o[a] + d[a] * (this_tmax - eps);
                cur[a] = clampi((int)((rs[a] - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
                fin[a] = clampi((int)((re[a] - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
                int start_index = cur[a] + (d[a] > 0 ? 1 : 0);
                float tmax_a = ((aabb[a] + ((float)start_index * vs[a] - rs[a])) / d[a]) + this_tmin;
                float sf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
                tdist[a] = (d[a] == 0.0f) ? this_tmax : tmax_a;
                stp[a] = (int)sf;
                delta[a] = (d[a] == 0.0f) ? this_tmax : vs[a] / d[a] * sf;
                ovf[a] = fin[a] + stp[a];
            }
            for (;;) {
                float t_traverse = fminf(tdist[0], fminf(tdist[1], tdist[2]));
                t_traverse = fminf(t_traverse, this_tmax);
                int64_t cell = ((int64_t)cur[0] * res[1] + cur[1]) * res[2] + cur[2];
                if (!binaries[cell]) {
                    if (step_size <= 0.0f) t_last = t_traverse;
                    else for (;;) {
                        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_traverse) break;
                        t_last += dt;
                    }
                    continuous = 0;
                } else {
                    for (;;) {
                        float t_next;
                        if (step_size <= 0.0f) t_next = t_traverse;
                        else {
                            float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                            if (t_last + dt * 0.5f >= t_traverse) break;
                            t_next = t_last + dt;
                        }
                        if (!continuous) {
                            if (!first_pass) {
                                int64_t idx = iv_base + n_intervals;
                                iv_vals[idx] = t_last; iv_ray[idx] = tid; iv_is_left[idx] = 1;
                            }
                            n_intervals++;
                            if (!first_pass) {
                                int64_t idx = iv_base + n_intervals;
                                iv_vals[idx] = t_next; iv_ray[idx] = tid; iv_is_right[idx] = 1;
                            }
                            n_intervals++;
                        } else {
                            if (!first_pass) {
                                int64_t idx = iv_base + n_intervals;
                                iv_vals[idx] = t_next; iv_ray[idx] = tid;
                                iv_is_left[idx - 1] = 1; iv_is_right[idx] = 1;
                            }
                            n_intervals++;
                        }
                        if (!first_pass) {
                            int64_t idx = sm_base + n_samples;
                            sm_vals[idx] = (t_next + t_last) * 0.5f; sm_ray[idx] = tid;
                        }
                        n_samples++;
                        continuous = 1;
                        t_last = t_next;
                        if (t_next >= t_traverse) break;
                    }
                }
                /* single_traversal: move to next voxel */
                int a;
                if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) a = 0;
                else if (tdist[1] < tdist[2]) a = 1;
                else a = 2;
                cur[a] += stp[a];
                tdist[a] += delta[a];
                if (cur[a] == ovf[a]) break;
            }
        }
    }
    if (first_pass) { iv_cnt[tid] = n_intervals; sm_cnt[tid] = n_samples; }
    if (term_plane) term_plane[tid] = t_last;
}

/* pass 1: per-ray counts of interval edges and samples */
IA_API int ia_ref_traverse_grids_count(
    int64_t n_rays, const float *rays_o, const float *rays_d,
    const int *res, const uint8_t *binaries, const float *aabb,
    const float *near_planes, const float *far_planes, float step_size, float cone_angle,
    int64_t *iv_cnt, int64_t *sm_cnt)
{
    for (int64_t i = 0; i < n_rays; i++)
        traverse_one_ray(i, rays_o, rays_d, res, binaries, aabb, near_planes[i], far_planes[i],
                         step_size, cone_angle, 1, 0, 0, iv_cnt, sm_cnt,
                         NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    return 0;
}

/* pass 2: fill.  iv_start / sm_start = exclusive scans of the counts.  The flag
 * arrays must be zero-initialised by the caller. */
IA_API int ia_ref_traverse_grids_fill(
    int64_t n_rays, const float *rays_o, const float *rays_d,
    const int *res, const uint8_t *binaries, const float *aabb,
    const float *near_planes, const float *far_planes, float step_size, float cone_angle,
    const int64_t *iv_start, const int64_t *sm_start,
    float *iv_vals, uint8_t *iv_is_left, uint8_t *iv_is_right, int64_t *iv_ray,
    float *sm_vals, int64_t *sm_ray, float *term_planes)
{
    for (int64_t i = 0; i < n_rays; i++)
        traverse_one_ray(i, rays_o, rays_d, res, binaries, aabb, near_planes[i], far_planes[i],
                         step_size, cone_angle, 0, iv_start[i], sm_start[i], NULL, NULL,
                         iv_vals, iv_is_left, iv_is_right, iv_ray, sm_vals, sm_ray, term_planes);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* T2: nerfacc.render_weight_from_alpha (call sites models/intrinsic_avatar.py:506,1199;
 *     models/volrend.py:162,764,952).  T_i = prod_{j<i in ray}(1-a_j), w_i = T_i a_i.
 *     Product is taken left-to-right (this file defines the order). */
IA_API int ia_ref_render_weight_from_alpha(
    int64_t n_rays, const int64_t *packed_info /*[n,2]*/, const float *alphas,
    float *weights, float *trans)
{
    for (int64_t r = 0; r < n_rays; r++) {
        int64_t b = packed_info[2 * r], n = packed_info[2 * r + 1];
        float T = 1.0f;
        for (int64_t j = 0; j < n; j++) {
            float a = alphas[b + j];
            trans[b + j] = T;
            weights[b + j] = T * a;
            T = T * (1.0f - a);
        }
    }
    return 0;
}

/* backward of T2 (closed form):
 *   d a_i = gw_i T_i - S_i / (1 - a_i),  S_i = sum_{j>i} (gw_j w_j + gT_j T_j)  */
IA_API int ia_ref_render_weight_from_alpha_bwd(
    int64_t n_rays, const int64_t *packed_info, const float *alphas,
    const float *weights, const float *trans, const float *g_weights, const float *g_trans,
    float *g_alphas)
{
    for (int64_t r = 0; r < n_rays; r++) {
        int64_t b = packed_info[2 * r], n = packed_info[2 * r + 1];
        float S = 0.0f;
        for (int64_t j = n - 1; j >= 0; j--) {
            int64_t i = b + j;
            float gw = g_weights ? g_weights[i] : 0.0f, gT = g_trans ? g_trans[i] : 0.0f;
            g_alphas[i] = gw * trans[i] - S / (1.0f - alphas[i]);
            S = S + (gw * weights[i] + gT * trans[i]);
        }
    }
    return 0;
}

/* T3: nerfacc.accumulate_along_rays (call sites models/volrend.py:176-187,783-797;
 *     models/intrinsic_avatar.py:1427-1453).  out[r,:] = sum_i w_i v_i, in sample order. */
IA_API int ia_ref_accumulate_along_rays(
    int64_t n_samples, int64_t n_rays, int dim, const float *weights,
    const float *values /* may be NULL => dim 1, v=1 */, const int64_t *ray_indices, float *out)
{
    memset(out, 0, sizeof(float) * (size_t)n_rays * (size_t)dim);
    for (int64_t i = 0; i < n_samples; i++) {
        int64_t r = ray_indices[i];
        for (int k = 0; k < dim; k++)
            out[r * dim + k] = out[r * dim + k] + (values ? weights[i] * values[i * dim + k] : weights[i]);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K5..K7 + pack_info: lib/nerfacc/cuda/csrc/pack.cu:7-82, lib/nerfacc/pack.py:46-77 */

IA_API int ia_ref_pack_info(int64_t n_samples, const int64_t *ray_indices, int64_t n_rays, int32_t *packed_info)
{
    /* pack.py:68-72: scatter_add of ones, cumsum, stack([cum - num, num]) */
    int32_t *num = (int32_t *)calloc((size_t)(n_rays > 0 ? n_rays : 1), sizeof(int32_t));
    for (int64_t i = 0; i < n_samples; i++) num[ray_indices[i]] += 1;
    int32_t cum = 0;
    for (int64_t r = 0; r < n_rays; r++) {
        cum += num[r];
        packed_info[2 * r] = cum - num[r];
        packed_info[2 * r + 1] = num[r];
    }
    free(num);
    return 0;
}

IA_API int ia_ref_unpack_info(int64_t n_rays, const int32_t *packed_info, int64_t *ray_indices)
{
    /* pack.cu:7-28 */
    for (int64_t i = 0; i < n_rays; i++) {
        int base = packed_info[2 * i], steps = packed_info[2 * i + 1];
        for (int j = 0; j < steps; j++) ray_indices[base + j] = i;
    }
    return 0;
}

IA_API int ia_ref_unpack_info_to_mask(int64_t n_rays, const int32_t *packed_info, int n_samples, uint8_t *masks)
{
    /* pack.cu:30-52; masks pre-zeroed [n_rays, n_samples] */
    memset(masks, 0, (size_t)n_rays * (size_t)n_samples);
    for (int64_t i = 0; i < n_rays; i++) {
        int steps = packed_info[2 * i + 1];
        for (int j = 0; j < steps; j++) masks[i * n_samples + j] = 1;
    }
    return 0;
}

IA_API int ia_ref_unpack_data(int64_t n_rays, const int32_t *packed_info, int data_dim,
                              const float *data, int n_samples_per_ray, float *out)
{
    /* pack.cu:55-82; out zero-filled [n_rays, n_samples_per_ray, data_dim] */
    memset(out, 0, sizeof(float) * (size_t)n_rays * (size_t)n_samples_per_ray * (size_t)data_dim);
    for (int64_t i = 0; i < n_rays; i++) {
        int base = packed_info[2 * i], steps = packed_info[2 * i + 1];
        for (int j = 0; j < steps; j++)
            for (int k = 0; k < data_dim; k++)
                out[((int64_t)i * n_samples_per_ray + j) * data_dim + k] = data[((int64_t)base + j) * data_dim + k];
    }
    return 0;
}

/* common prologue of K1..K4 (cdf.cu:177-183): resample_packed_info */
IA_API int64_t ia_ref_resample_packed_info(int64_t n_rays, const int32_t *packed_info, int n, int add_steps,
                                           int32_t *resample_packed_info)
{
    int32_t cum = 0;
    for (int64_t r = 0; r < n_rays; r++) {
        int32_t steps = packed_info[2 * r + 1];
        int32_t cnt = (steps > 0 ? n : 0) + (add_steps ? steps : 0);
        resample_packed_info[2 * r] = cum;
        resample_packed_info[2 * r + 1] = cnt;
        cum += cnt;
    }
    return cum;
}

/* K1: cdf_resampling_kernel, lib/nerfacc/cuda/csrc/cdf.cu:10-149.
 * ts/offsets/indices are fully written for every ray with steps>0; fg_counts and
 * bg_counts must be zeroed and surface_idx set to -1 by the caller (cdf.cu:189-191). */
IA_API int ia_ref_ray_resampling(
    int64_t n_rays, const int32_t *packed_info, const float *starts, const float *ends,
    const float *weights_all, const float *sdfs_all, const int32_t *resample_packed_info,
    float *resample_ts, float *resample_offsets, int64_t *surface_idx,
    int64_t *resample_indices, int32_t *resample_fg_counts, int32_t *resample_bg_counts)
{
    for (int64_t i = 0; i < n_rays; i++) {
        const int base = packed_info[i * 2 + 0], steps = packed_info[i * 2 + 1];
        const int rbase = resample_packed_info[i * 2 + 0], rsteps = resample_packed_info[i * 2 + 1];
        if (steps == 0) continue;
        const float *st = starts + base, *en = ends + base, *w = weights_all + base, *sdfs = sdfs_all + base;
        int32_t *fgc = resample_fg_counts + base;
        float *ts = resample_ts + rbase, *offs = resample_offsets + rbase;
        int64_t *idxs = resample_indices + rbase;

        float weights_sum = 0.0f;
        for (int j = 0; j < steps; j++) weights_sum += w[j];
        weights_sum += fmaxf(1.0f - weights_sum, 0.0f);

        int num_bins = rsteps;
        float cdf_step_size = (float)((1.0f - 1.0 / num_bins) / (rsteps - 1));
        int idx = 0, j = 0;
        float cdf_prev = 0.0f, cdf_next = w[idx] / weights_sum;
        float cdf_u = (float)(1.0 / (2 * num_bins));
        float sdf_prev = sdfs[0];
        float sdf_next = 0.0f;
        if (steps > 1) sdf_next = sdfs[1];
        int found_surface = 0;
        while (j < num_bins && idx < steps) {
            if (cdf_u < cdf_next) {
                float scaling = (en[idx] - st[idx]) / (cdf_next - cdf_prev);
                float offset = (cdf_u - cdf_prev) * scaling;
                float t = offset + st[idx];
                if (sdf_prev >= 0 && sdf_next < 0 && !found_surface) {
                    float sdf_approx = sdf_prev + (sdf_next - sdf_prev) * (offset / (en[idx] - st[idx]));
                    ts[j] = sdf_approx >= 0 ? t : (j > 0 ? ts[j - 1] : st[idx]);
                } else if (found_surface) {
                    ts[j] = j > 0 ? ts[j - 1] : st[idx];
                } else {
                    ts[j] = t;
                }
                offs[j] = offset;
                idxs[j] = idx + base;
                fgc[idx] += 1;
                cdf_u += cdf_step_size;
                j += 1;
            } else if (idx < steps - 1) {
                idx += 1;
                if (sdf_prev >= 0 && sdf_next < 0 && !found_surface) {
                    surface_idx[i] = idx - 1 + base;
                    found_surface = 1;
                }
                sdf_prev = sdfs[idx];
                sdf_next = idx < steps - 1 ? sdfs[idx + 1] : 0.0f;
                cdf_prev = cdf_next;
                cdf_next += w[idx] / weights_sum;
            } else {
                break;
            }
        }
        while (j < num_bins) {
            float offset = 10000.f;
            float t = offset + en[steps - 1];
            ts[j] = t;
            offs[j] = offset;
            idxs[j] = steps - 1 + base;
            cdf_u += cdf_step_size;
            j += 1;
            resample_bg_counts[i] += 1;
        }
    }
    return 0;
}

/* K2: cdf_resampling_merge_kernel, cdf.cu:217-334.  All outputs zero-initialised
 * by the caller (cdf.cu:372-377). */
IA_API int ia_ref_ray_resampling_merge(
    int64_t n_rays, const int32_t *packed_info, const float *vals_all,
    const uint8_t *is_left_all, const uint8_t *is_right_all, const float *weights_all,
    const int32_t *resample_packed_info,
    float *resample_vals, float *resample_dists, uint8_t *resample_is_left,
    uint8_t *resample_is_right, uint8_t *is_resample, uint8_t *is_fg_sample)
{
    for (int64_t i = 0; i < n_rays; i++) {
        const int base = packed_info[i * 2 + 0], steps = packed_info[i * 2 + 1];
        const int rbase = resample_packed_info[i * 2 + 0];
        const int rsteps = resample_packed_info[i * 2 + 1] - steps;
        if (steps == 0) continue;
        const float *vals = vals_all + base, *w = weights_all + base;
        const uint8_t *il = is_left_all + base, *ir = is_right_all + base;
        uint8_t *fg = is_fg_sample + rbase, *ol = resample_is_left + rbase, *orr = resample_is_right + rbase,
                *ors = is_resample + rbase;
        float *ov = resample_vals + rbase, *od = resample_dists + rbase;

        float weights_sum = 0.0f;
        for (int j = 0; j < steps - 1; j++) weights_sum += (il[j] && ir[j + 1]) ? w[j] : 0.0f;
        weights_sum += fmaxf(1.0f - weights_sum, 0.0f);

        int num_bins = rsteps;
        float cdf_step_size = (float)((1.0f - 1.0 / num_bins) / (rsteps - 1));
        int idx = 0, j = 0;
        float start = 0.0f, end = 0.0f;
        float cdf_prev = 0.0f, cdf_next = w[idx] / weights_sum;
        float cdf_u = (float)(1.0 / (2 * num_bins));
        start = vals[0];
        end = vals[1];
        ov[0] = start;
        fg[0] = 1;
        ol[0] = 1;
        while (j < num_bins && idx < steps - 1) {
            if (cdf_u < cdf_next) {
                float scaling = (end - start) / (cdf_next - cdf_prev);
                float offset = (cdf_u - cdf_prev) * scaling;
                float t = offset + start;
                cdf_u += cdf_step_size;
                od[j + idx] = t - ov[j + idx];
                j += 1;
                ov[j + idx] = t;
                fg[j + idx] = 1;
                ors[j + idx] = 1;
                ol[j + idx] = 1;
                orr[j + idx] = 1;
            } else {
                od[j + idx] = end - ov[j + idx];
                idx += 1;
                ov[j + idx] = end;
                fg[j + idx] = 1;
                orr[j + idx] = ir[idx];
                if (idx >= steps - 1) break;
                start = vals[idx];
                end = vals[idx + 1];
                if (il[idx] && ir[idx + 1]) {
                    cdf_prev = cdf_next;
                    cdf_next += w[idx] / weights_sum;
                    ol[j + idx] = 1;
                }
            }
        }
        while (idx < steps - 1) {
            od[j + idx] = end - ov[j + idx];
            idx += 1;
            ov[j + idx] = end;
            fg[j + idx] = 1;
            orr[j + idx] = ir[idx];
            if (idx >= steps - 1) break;
            start = vals[idx];
            end = vals[idx + 1];
            if (il[idx] && ir[idx + 1]) ol[j + idx] = 1;
        }
    }
    return 0;
}

/* K3: cdf_resampling_fine_kernel, cdf.cu:403-478 (outputs zero-initialised, cdf.cu:512-514) */
IA_API int ia_ref_ray_resampling_fine(
    int64_t n_rays, const int32_t *packed_info, const float *starts, const float *ends,
    const float *weights_all, const int32_t *resample_packed_info,
    float *resample_starts, float *resample_ends, uint8_t *is_fg_sample)
{
    for (int64_t i = 0; i < n_rays; i++) {
        const int base = packed_info[i * 2 + 0], steps = packed_info[i * 2 + 1];
        const int rbase = resample_packed_info[i * 2 + 0], rsteps = resample_packed_info[i * 2 + 1];
        if (steps == 0) continue;
        const float *st = starts + base, *en = ends + base, *w = weights_all + base;
        float *os = resample_starts + rbase, *oe = resample_ends + rbase;
        uint8_t *fg = is_fg_sample + rbase;

        float weights_sum = 0.0f;
        for (int j = 0; j < steps; j++) weights_sum += w[j];
        weights_sum += fmaxf(1.0f - weights_sum, 0.0f);

        int num_bins = rsteps + 1;
        float cdf_step_size = (float)((1.0f - 1.0 / num_bins) / rsteps);
        int idx = 0, j = 0;
        float cdf_prev = 0.0f, cdf_next = w[idx] / weights_sum;
        float cdf_u = (float)(1.0 / (2 * num_bins));
        while (j < num_bins && idx < steps) {
            if (cdf_u < cdf_next) {
                float scaling = (en[idx] - st[idx]) / (cdf_next - cdf_prev);
                float t = (cdf_u - cdf_prev) * scaling + st[idx];
                if (j < num_bins - 1) os[j] = t;
                if (j > 0) { oe[j - 1] = t; fg[j - 1] = 1; }
                cdf_u += cdf_step_size;
                j += 1;
            } else {
                idx += 1;
                if (idx >= steps) break;
                cdf_prev = cdf_next;
                cdf_next += w[idx] / weights_sum;
            }
        }
    }
    return 0;
}

/* K4: cdf_resampling_sdf_fine_kernel, cdf.cu:536-638 */
IA_API int ia_ref_ray_resampling_sdf_fine(
    int64_t n_rays, const int32_t *packed_info, const float *starts, const float *ends,
    const float *alphas_all, const float *sdfs_all, const int32_t *resample_packed_info,
    float *resample_starts, float *resample_ends, uint8_t *is_fg_sample)
{
    for (int64_t i = 0; i < n_rays; i++) {
        const int base = packed_info[i * 2 + 0], steps = packed_info[i * 2 + 1];
        const int rbase = resample_packed_info[i * 2 + 0], rsteps = resample_packed_info[i * 2 + 1];
        if (steps == 0) continue;
        const float *st = starts + base, *en = ends + base, *al = alphas_all + base, *sdfs = sdfs_all + base;
        float *os = resample_starts + rbase, *oe = resample_ends + rbase;
        uint8_t *fg = is_fg_sample + rbase;

        int idx = 0;
        float sdf_prev = sdfs[0];
        int found_surface = 0;
        while (idx < steps) {
            idx += 1;
            if (idx >= steps) break;
            if (sdf_prev >= 0 && sdfs[idx] < 0 && !found_surface) {
                idx -= 1;
                found_surface = 1;
                break;
            }
            sdf_prev = sdfs[idx];
        }
        if (!found_surface) continue;

        int num_bins = rsteps + 1;
        float cdf_step_size = (float)((1.0f - 1.0 / num_bins) / rsteps);
        int j = 0;
        float trans = 1.0f;
        float weight = al[idx];
        trans *= (1.0f - al[idx]);
        float cdf_prev = 0.0f, cdf_next = weight;
        float cdf_u = (float)(1.0 / (2 * num_bins));
        while (j < num_bins && idx < steps) {
            if (cdf_u < cdf_next) {
                float scaling = (en[idx] - st[idx]) / (cdf_next - cdf_prev);
                float t = (cdf_u - cdf_prev) * scaling + st[idx];
                if (j < num_bins - 1) os[j] = t;
                if (j > 0) { oe[j - 1] = t; fg[j - 1] = 1; }
                cdf_u += cdf_step_size;
                j += 1;
            } else {
                idx += 1;
                if (idx >= steps) break;
                weight = trans * al[idx];
                trans *= (1.0f - al[idx]);
                cdf_prev = cdf_next;
                cdf_next += weight;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* fast-SNARF kernels: models/deformers/fast_snarf/cuda/... */

/* K10: precompute_kernel, precompute/precompute.cu:24-71.
 * voxel_w [1,24,D,H,W], tfs [B,24,4,4] -> voxel_d [B,3,D,H,W], voxel_J [B,12,D,H,W] */
IA_API int ia_ref_precompute(
    int B, int D, int H, int W, const float *voxel_w, const float *tfs,
    const float *offset /*[3]*/, const float *scale /*[3]*/, float *voxel_d, float *voxel_J)
{
    const int64_t vol = (int64_t)D * H * W;
    for (int64_t index = 0; index < (int64_t)B * vol; index++) {
        int idx_b = (int)(index / vol);
        int idx_d = (int)(index % vol / ((int64_t)H * W));
        int idx_h = (int)(index % vol % ((int64_t)H * W) / W);
        int idx_w = (int)(index % vol % ((int64_t)H * W) % W);
        float coord_x = (((float)idx_w) / (W - 1) * 2 - 1) / scale[0] - offset[0];
        float coord_y = (((float)idx_h) / (H - 1) * 2 - 1) / scale[1] - offset[1];
        float coord_z = (((float)idx_d) / (D - 1) * 2 - 1) / scale[2] - offset[2];
        float J[12];
        int64_t v = ((int64_t)idx_d * H + idx_h) * W + idx_w;
        for (int i0 = 0; i0 < 3; i0++)
            for (int i1 = 0; i1 < 4; i1++) {
                J[i0 * 4 + i1] = 0;
                for (int j = 0; j < 24; j++)
                    J[i0 * 4 + i1] += voxel_w[j * vol + v] * tfs[((idx_b * 24 + j) * 4 + i0) * 4 + i1];
            }
        for (int c = 0; c < 12; c++) voxel_J[((int64_t)idx_b * 12 + c) * vol + v] = J[c];
        for (int i0 = 0; i0 < 3; i0++) {
            float xi = J[i0 * 4 + 0] * coord_x + J[i0 * 4 + 1] * coord_y + J[i0 * 4 + 2] * coord_z + J[i0 * 4 + 3];
            voxel_d[((int64_t)idx_b * 3 + i0) * vol + v] = xi;
        }
    }
    return 0;
}

/* trilinear 12-channel fetch, align_corners=true, zero outside:
 * fuse_cuda_kernel_fast.cu:62-233 (grid_sampler_3d, !nearest branch) */
static void grid_sample_J(const float *vJ /*[12,D,H,W] of this batch*/, int D, int H, int W,
                          float gx, float gy, float gz, float *out /*12*/)
{
    const int64_t vol = (int64_t)D * H * W;
    float ix = ((gx + 1.f) / 2) * (W - 1);
    float iy = ((gy + 1.f) / 2) * (H - 1);
    float iz = ((gz + 1.f) / 2) * (D - 1);
    /* safe_downgrade_to_int_range, :83-91 */
    if (ix > 2147483646.0f || ix < -2147483648.0f || !isfinite((double)ix)) ix = -100.0f;
    if (iy > 2147483646.0f || iy < -2147483648.0f || !isfinite((double)iy)) iy = -100.0f;
    if (iz > 2147483646.0f || iz < -2147483648.0f || !isfinite((double)iz)) iz = -100.0f;
    int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    float tnw = (x1 - ix) * (y1 - iy) * (z1 - iz);
    float tne = (ix - x0) * (y1 - iy) * (z1 - iz);
    float tsw = (x1 - ix) * (iy - y0) * (z1 - iz);
    float tse = (ix - x0) * (iy - y0) * (z1 - iz);
    float bnw = (x1 - ix) * (y1 - iy) * (iz - z0);
    float bne = (ix - x0) * (y1 - iy) * (iz - z0);
    float bsw = (x1 - ix) * (iy - y0) * (iz - z0);
    float bse = (ix - x0) * (iy - y0) * (iz - z0);
#define INB(z, y, x) ((z) >= 0 && (z) < D && (y) >= 0 && (y) < H && (x) >= 0 && (x) < W)
#define AT(c, z, y, x) vJ[(c) * vol + ((int64_t)(z) * H + (y)) * W + (x)]
    for (int c = 0; c < 12; c++) {
        float o = 0;
        if (INB(z0, y0, x0)) o += AT(c, z0, y0, x0) * tnw;
        if (INB(z0, y0, x1)) o += AT(c, z0, y0, x1) * tne;
        if (INB(z0, y1, x0)) o += AT(c, z0, y1, x0) * tsw;
        if (INB(z0, y1, x1)) o += AT(c, z0, y1, x1) * tse;
        if (INB(z1, y0, x0)) o += AT(c, z1, y0, x0) * bnw;
        if (INB(z1, y0, x1)) o += AT(c, z1, y0, x1) * bne;
        if (INB(z1, y1, x0)) o += AT(c, z1, y1, x0) * bsw;
        if (INB(z1, y1, x1)) o += AT(c, z1, y1, x1) * bse;
        out[c] = o;
    }
#undef INB
#undef AT
}

/* fuse_J_inv_update, fuse_cuda_kernel_fast.cu:22-55 */
static void J_inv_update(float *Ji, float x0, float x1, float x2, float g0, float g1, float g2)
{
    float J00 = Ji[0], J01 = Ji[1], J02 = Ji[2], J10 = Ji[3], J11 = Ji[4], J12 = Ji[5], J20 = Ji[6], J21 = Ji[7],
          J22 = Ji[8];
    float c0 = J00 * x0 + J10 * x1 + J20 * x2;
    float c1 = J01 * x0 + J11 * x1 + J21 * x2;
    float c2 = J02 * x0 + J12 * x1 + J22 * x2;
    float s = c0 * g0 + c1 * g1 + c2 * g2;
    float r0 = -J00 * g0 - J01 * g1 - J02 * g2;
    float r1 = -J10 * g0 - J11 * g1 - J12 * g2;
    float r2 = -J20 * g0 - J21 * g1 - J22 * g2;
    Ji[0] += c0 * (r0 + x0) / s;
    Ji[1] += c1 * (r0 + x0) / s;
    Ji[2] += c2 * (r0 + x0) / s;
    Ji[3] += c0 * (r1 + x1) / s;
    Ji[4] += c1 * (r1 + x1) / s;
    Ji[5] += c2 * (r1 + x1) / s;
    Ji[6] += c0 * (r2 + x2) / s;
    Ji[7] += c1 * (r2 + x2) / s;
    Ji[8] += c2 * (r2 + x2) / s;
}

/* K8: broyden_kernel, fuse_cuda_kernel_fast.cu:250-413.
 * x [B,N,I,3], J_inv [B,N,I,3,3], is_valid [B,N,I] are caller-zeroed outputs
 * (deformer_torch.py:113-115); only converged+inside candidates are written. */
IA_API int ia_ref_fuse_broyden(
    int B, int64_t N, int I, const float *xd_tgt /*[B,N,3]*/, const float *voxel_J /*[B,12,D,H,W]*/,
    int D, int H, int W, const float *tfs /*[B,24,4,4]*/, const int32_t *bone_ids /*[I]*/,
    const float *offset /*[3]*/, const float *scale /*[3]*/, float cvg_threshold, float dvg_threshold,
    float *x, float *J_inv, uint8_t *is_valid)
{
    const int64_t vol = (int64_t)D * H * W;
    for (int64_t index = 0; index < (int64_t)B * N * I; index++) {
        const int i_batch = (int)(index / (N * I));
        const int64_t i_point = (index % (N * I)) / I;
        const int i_init = (int)((index % (N * I)) % I);
        const float *vJ = voxel_J + (int64_t)i_batch * 12 * vol;
        float gx[3], gx_new[3], xt[3], x_l[3];
        xt[0] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 0];
        xt[1] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 1];
        xt[2] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 2];
        int i_bone = bone_ids[i_init];
        const float *T = tfs + ((int64_t)i_batch * 24 + i_bone) * 16;
        float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
        x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
        x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
        x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];

        float Jl[12];
        grid_sample_J(vJ, D, H, W, scale[0] * (x_l[0] + offset[0]), scale[1] * (x_l[1] + offset[1]),
                      scale[2] * (x_l[2] + offset[2]), Jl);
        float Ji[9];
        Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
        Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
        Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];

        for (int it = 0; it < 10; it++) {
            float J00 = Ji[0], J01 = Ji[1], J02 = Ji[2], J10 = Ji[3], J11 = Ji[4], J12 = Ji[5], J20 = Ji[6],
                  J21 = Ji[7], J22 = Ji[8];
            if (it == 0) {
                gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3];
                gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7];
                gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11];
                gx[0] = gx[0] - xt[0]; gx[1] = gx[1] - xt[1]; gx[2] = gx[2] - xt[2];
            } else {
                gx[0] = gx_new[0]; gx[1] = gx_new[1]; gx[2] = gx_new[2];
            }
            float u0 = -J00 * gx[0] + -J01 * gx[1] + -J02 * gx[2];
            float u1 = -J10 * gx[0] + -J11 * gx[1] + -J12 * gx[2];
            float u2 = -J20 * gx[0] + -J21 * gx[1] + -J22 * gx[2];
            x_l[0] += u0; x_l[1] += u1; x_l[2] += u2;
            float ix = scale[0] * (x_l[0] + offset[0]);
            float iy = scale[1] * (x_l[1] + offset[1]);
            float iz = scale[2] * (x_l[2] + offset[2]);
            grid_sample_J(vJ, D, H, W, ix, iy, iz, Jl);
            gx_new[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
            gx_new[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
            gx_new[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
            float norm_gx = gx_new[0] * gx_new[0] + gx_new[1] * gx_new[1] + gx_new[2] * gx_new[2];
            if (norm_gx < cvg_threshold * cvg_threshold) {
                int ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
                is_valid[index] = (uint8_t)ok;
                if (ok) {
                    x[index * 3 + 0] = x_l[0]; x[index * 3 + 1] = x_l[1]; x[index * 3 + 2] = x_l[2];
                    float *Jo = J_inv + index * 9;
                    Jo[0] = J00; Jo[1] = J01; Jo[2] = J02; Jo[3] = J10; Jo[4] = J11; Jo[5] = J12;
                    Jo[6] = J20; Jo[7] = J21; Jo[8] = J22;
                }
                break;
            } else if (norm_gx > dvg_threshold * dvg_threshold) {
                is_valid[index] = 0;
                break;
            }
            J_inv_update(Ji, u0, u1, u2, gx_new[0] - gx[0], gx_new[1] - gx[1], gx_new[2] - gx[2]);
        }
    }
    return 0;
}

/* K9: filter, filter/filter.cu:10-54 (B == 1 on this path) */
IA_API int ia_ref_filter(int64_t N, int I, const float *x /*[N,I,3]*/, const uint8_t *mask, uint8_t *out)
{
    for (int64_t p = 0; p < N; p++) {
        for (int i = 0; i < I; i++) {
            if (!mask[p * I + i]) { out[p * I + i] = 0; continue; }
            float xi0 = x[(p * I + i) * 3 + 0], xi1 = x[(p * I + i) * 3 + 1], xi2 = x[(p * I + i) * 3 + 2];
            int flag = 1;
            for (int j = i + 1; j < I; j++) {
                if (!mask[p * I + j]) continue;
                float d0 = xi0 - x[(p * I + j) * 3 + 0];
                float d1 = xi1 - x[(p * I + j) * 3 + 1];
                float d2 = xi2 - x[(p * I + j) * 3 + 2];
                float dist = d0 * d0 + d1 * d1 + d2 * d2;
                if (dist < 0.0001 * 0.0001) { flag = 0; break; }   /* double compare, as in filter.cu:43 */
            }
            out[p * I + i] = (uint8_t)flag;
        }
    }
    return 0;
}
