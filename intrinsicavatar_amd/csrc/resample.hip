// resample.hip -- CDF importance resampling + pack/unpack for gfx950.
// Replaces the reference's customised nerfacc kernels:
//   K1 cdf_resampling_kernel            lib/nerfacc/cuda/csrc/cdf.cu:10-149
//   K2 cdf_resampling_merge_kernel      cdf.cu:217-334
//   K3 cdf_resampling_fine_kernel       cdf.cu:403-478
//   K4 cdf_resampling_sdf_fine_kernel   cdf.cu:536-638
//   K5 unpack_info_kernel, K6 unpack_info_to_mask_kernel, K7 unpack_data_kernel  pack.cu:7-82
//   pack_info (torch ops)               lib/nerfacc/pack.py:46-77
// and the host prologue of K1..K4 (cdf.cu:177-183) without the two .item() syncs.
//
// Every CDF walk is inherently serial per ray; rays are independent => one ray per lane.
// The float expression order matches oracle/ia_oracle.c exactly (TU built with
// -ffp-contract=off), so the integer outputs (indices, counts, flags) are bit-exact.
#include "ia_common.h"

namespace {

constexpr int THREADS = 256;

// ---- prologue: counts + scan ------------------------------------------------
__global__ __launch_bounds__(THREADS) void resample_counts_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                                   int n, int add_steps, int32_t* __restrict__ cnt)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int steps = packed_info[2 * r + 1];
    cnt[r] = (steps > 0 ? n : 0) + (add_steps ? steps : 0);
}

__global__ __launch_bounds__(THREADS) void stack_info_kernel(int64_t n_rays, const int32_t* __restrict__ start,
                                                              const int32_t* __restrict__ cnt, int32_t* __restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    reinterpret_cast<int2*>(out)[r] = make_int2(start[r], cnt[r]);
}

// samples per ray (scatter_add of ones, lib/nerfacc/pack.py:70-74).  Integer atomics: exact for any input order; for the
// usual ray-sorted input the wave first merges runs of equal indices (ballot) and issues one atomic per run.
__global__ __launch_bounds__(THREADS) void count_rays_kernel(int64_t n_samples, const int64_t* __restrict__ ray_indices,
                                                              int32_t* __restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = i < n_samples;
    const int64_t r = active ? ray_indices[i] : -1;
    const int64_t prev = __shfl_up(r, 1, 64);
    const bool head = (lane == 0) || (prev != r);
    const unsigned long long heads = __ballot(head);
    if (head && active) {
        const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
        const int len = above ? (__builtin_ctzll(above) + 1) : (64 - lane);
        atomicAdd(&cnt[r], len);
    }
}

// ---- K5..K7 -------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void unpack_info_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                               int64_t* __restrict__ ray_indices)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n_rays) return;
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[i];
    for (int j = 0; j < pi.y; ++j) ray_indices[pi.x + j] = i;
}

__global__ __launch_bounds__(THREADS) void unpack_mask_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                               int n_samples, uint8_t* __restrict__ masks)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n_rays) return;
    const int steps = packed_info[2 * i + 1];
    for (int j = 0; j < steps; ++j) masks[i * n_samples + j] = 1;
}

__global__ __launch_bounds__(THREADS) void unpack_data_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                               int data_dim, const float* __restrict__ data,
                                                               int n_per_ray, float* __restrict__ out)
{
    // one lane per (ray, slot): coalesced on the dense side
    const int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (t >= n_rays * n_per_ray) return;
    const int64_t r = t / n_per_ray;
    const int j = (int)(t % n_per_ray);
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    if (j >= pi.y) return;
    for (int k = 0; k < data_dim; k++) out[t * data_dim + k] = data[(int64_t)(pi.x + j) * data_dim + k];
}

// ---- K1 -----------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void k1_resampling_kernel(
    int64_t n_rays, const int32_t* __restrict__ packed_info, const float* __restrict__ starts,
    const float* __restrict__ ends, const float* __restrict__ weights_all, const float* __restrict__ sdfs_all,
    const int32_t* __restrict__ resample_packed_info, float* __restrict__ resample_ts,
    float* __restrict__ resample_offsets, int64_t* __restrict__ surface_idx, int64_t* __restrict__ resample_indices,
    int32_t* __restrict__ resample_fg_counts, int32_t* __restrict__ resample_bg_counts)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n_rays) return;
    const int base = packed_info[i * 2 + 0], steps = packed_info[i * 2 + 1];
    const int rbase = resample_packed_info[i * 2 + 0], rsteps = resample_packed_info[i * 2 + 1];
    if (steps == 0) return;
    const float *st = starts + base, *en = ends + base, *w = weights_all + base, *sdfs = sdfs_all + base;
    int32_t* fgc = resample_fg_counts + base;
    float *ts = resample_ts + rbase, *offs = resample_offsets + rbase;
    int64_t* idxs = resample_indices + rbase;

    float weights_sum = 0.0f;
    for (int j = 0; j < steps; j++) weights_sum += w[j];
    weights_sum += fmaxf(1.0f - weights_sum, 0.0f);

    const int num_bins = rsteps;
    const float cdf_step_size = (float)((1.0f - 1.0 / num_bins) / (rsteps - 1));
    int idx = 0, j = 0;
    float cdf_prev = 0.0f, cdf_next = w[idx] / weights_sum;
    float cdf_u = (float)(1.0 / (2 * num_bins));
    float sdf_prev = sdfs[0];
    float sdf_next = 0.0f;
    if (steps > 1) sdf_next = sdfs[1];
    bool found_surface = false;
    float t_prev = 0.0f;   // == ts[j-1] (kept in a register instead of re-reading HBM)
    int fg_here = 0;       // pending fg count of interval idx
    float st_i = st[0], en_i = en[0];
    int bg = 0;
    while (j < num_bins && idx < steps) {
        if (cdf_u < cdf_next) {
            const float scaling = (en_i - st_i) / (cdf_next - cdf_prev);
            const float offset = (cdf_u - cdf_prev) * scaling;
            const float t = offset + st_i;
            float tv;
            if (sdf_prev >= 0 && sdf_next < 0 && !found_surface) {
                const float sdf_approx = sdf_prev + (sdf_next - sdf_prev) * (offset / (en_i - st_i));
                tv = sdf_approx >= 0 ? t : (j > 0 ? t_prev : st_i);
            } else if (found_surface) {
                tv = j > 0 ? t_prev : st_i;
            } else {
                tv = t;
            }
            ts[j] = tv;
            t_prev = tv;
            offs[j] = offset;
            idxs[j] = idx + base;
            fg_here += 1;
            cdf_u += cdf_step_size;
            j += 1;
        } else if (idx < steps - 1) {
            if (fg_here) { fgc[idx] = fg_here; fg_here = 0; }
            idx += 1;
            if (sdf_prev >= 0 && sdf_next < 0 && !found_surface) {
                surface_idx[i] = idx - 1 + base;
                found_surface = true;
            }
            sdf_prev = sdfs[idx];
            sdf_next = idx < steps - 1 ? sdfs[idx + 1] : 0.0f;
            cdf_prev = cdf_next;
            cdf_next += w[idx] / weights_sum;
            st_i = st[idx];
            en_i = en[idx];
        } else {
            break;
        }
    }
    if (fg_here) fgc[idx] = fg_here;
    const float en_last = en[steps - 1];
    while (j < num_bins) {
        const float offset = 10000.f;
        ts[j] = offset + en_last;
        offs[j] = offset;
        idxs[j] = steps - 1 + base;
        j += 1;
        bg += 1;
    }
    if (bg) resample_bg_counts[i] = bg;
}

// ---- K2 -----------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void k2_merge_kernel(
    int64_t n_rays, const int32_t* __restrict__ packed_info, const float* __restrict__ vals_all,
    const uint8_t* __restrict__ is_left_all, const uint8_t* __restrict__ is_right_all,
    const float* __restrict__ weights_all, const int32_t* __restrict__ resample_packed_info,
    float* __restrict__ resample_vals, float* __restrict__ resample_dists, uint8_t* __restrict__ resample_is_left,
    uint8_t* __restrict__ resample_is_right, uint8_t* __restrict__ is_resample, uint8_t* __restrict__ is_fg_sample)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n_rays) return;
    const int base = packed_info[i * 2 + 0], steps = packed_info[i * 2 + 1];
    const int rbase = resample_packed_info[i * 2 + 0];
    const int rsteps = resample_packed_info[i * 2 + 1] - steps;
    if (steps == 0) return;
    const float *vals = vals_all + base, *w = weights_all + base;
    const uint8_t *il = is_left_all + base, *ir = is_right_all + base;
    uint8_t *fg = is_fg_sample + rbase, *ol = resample_is_left + rbase, *orr = resample_is_right + rbase,
            *ors = is_resample + rbase;
    float *ov = resample_vals + rbase, *od = resample_dists + rbase;

    float weights_sum = 0.0f;
    for (int j = 0; j < steps - 1; j++) weights_sum += (il[j] && ir[j + 1]) ? w[j] : 0.0f;
    weights_sum += fmaxf(1.0f - weights_sum, 0.0f);

    const int num_bins = rsteps;
    const float cdf_step_size = (float)((1.0f - 1.0 / num_bins) / (rsteps - 1));
    int idx = 0, j = 0;
    float start = 0.0f, end = 0.0f;
    float cdf_prev = 0.0f, cdf_next = w[idx] / weights_sum;
    float cdf_u = (float)(1.0 / (2 * num_bins));
    start = vals[0];
    end = steps > 1 ? vals[1] : 0.0f;
    float v_last = start;   // == ov[j + idx] of the most recent write
    ov[0] = start;
    fg[0] = 1;
    ol[0] = 1;
    while (j < num_bins && idx < steps - 1) {
        if (cdf_u < cdf_next) {
            const float scaling = (end - start) / (cdf_next - cdf_prev);
            const float offset = (cdf_u - cdf_prev) * scaling;
            const float t = offset + start;
            cdf_u += cdf_step_size;
            od[j + idx] = t - v_last;
            j += 1;
            ov[j + idx] = t;
            v_last = t;
            fg[j + idx] = 1;
            ors[j + idx] = 1;
            ol[j + idx] = 1;
            orr[j + idx] = 1;
        } else {
            od[j + idx] = end - v_last;
            idx += 1;
            ov[j + idx] = end;
            v_last = end;
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
        od[j + idx] = end - v_last;
        idx += 1;
        ov[j + idx] = end;
        v_last = end;
        fg[j + idx] = 1;
        orr[j + idx] = ir[idx];
        if (idx >= steps - 1) break;
        start = vals[idx];
        end = vals[idx + 1];
        if (il[idx] && ir[idx + 1]) ol[j + idx] = 1;
    }
}

// ---- K3 / K4 --------------------------------------------------------------------
template <bool SDF>
__global__ __launch_bounds__(THREADS) void k34_fine_kernel(
    int64_t n_rays, const int32_t* __restrict__ packed_info, const float* __restrict__ starts,
    const float* __restrict__ ends, const float* __restrict__ wa_all /* weights (K3) or alphas (K4) */,
    const float* __restrict__ sdfs_all, const int32_t* __restrict__ resample_packed_info,
    float* __restrict__ resample_starts, float* __restrict__ resample_ends, uint8_t* __restrict__ is_fg_sample)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n_rays) return;
    const int base = packed_info[i * 2 + 0], steps = packed_info[i * 2 + 1];
    const int rbase = resample_packed_info[i * 2 + 0], rsteps = resample_packed_info[i * 2 + 1];
    if (steps == 0) return;
    const float *st = starts + base, *en = ends + base, *wa = wa_all + base;
    float *os = resample_starts + rbase, *oe = resample_ends + rbase;
    uint8_t* fg = is_fg_sample + rbase;

    int idx = 0;
    float weights_sum = 0.0f, trans = 1.0f, cdf_next;
    if (SDF) {
        const float* sdfs = sdfs_all + base;
        float sdf_prev = sdfs[0];
        bool found_surface = false;
        while (idx < steps) {
            idx += 1;
            if (idx >= steps) break;
            if (sdf_prev >= 0 && sdfs[idx] < 0 && !found_surface) {
                idx -= 1;
                found_surface = true;
                break;
            }
            sdf_prev = sdfs[idx];
        }
        if (!found_surface) return;
        const float weight = wa[idx];
        trans *= (1.0f - wa[idx]);
        cdf_next = weight;
    } else {
        for (int j = 0; j < steps; j++) weights_sum += wa[j];
        weights_sum += fmaxf(1.0f - weights_sum, 0.0f);
        cdf_next = wa[idx] / weights_sum;
    }
    const int num_bins = rsteps + 1;
    const float cdf_step_size = (float)((1.0f - 1.0 / num_bins) / rsteps);
    int j = 0;
    float cdf_prev = 0.0f;
    float cdf_u = (float)(1.0 / (2 * num_bins));
    while (j < num_bins && idx < steps) {
        if (cdf_u < cdf_next) {
            const float scaling = (en[idx] - st[idx]) / (cdf_next - cdf_prev);
            const float t = (cdf_u - cdf_prev) * scaling + st[idx];
            if (j < num_bins - 1) os[j] = t;
            if (j > 0) { oe[j - 1] = t; fg[j - 1] = 1; }
            cdf_u += cdf_step_size;
            j += 1;
        } else {
            idx += 1;
            if (idx >= steps) break;
            if (SDF) {
                const float weight = trans * wa[idx];
                trans *= (1.0f - wa[idx]);
                cdf_prev = cdf_next;
                cdf_next += weight;
            } else {
                cdf_prev = cdf_next;
                cdf_next += wa[idx] / weights_sum;
            }
        }
    }
}


// ---- foreground compaction of a fine re-sampling (K3 / K4 outputs) -----------------------------------------------------------
// The caller of ray_resampling_sdf_fine keeps the foreground intervals only (models/intrinsic_avatar.py:516-528: three
// boolean-mask gathers + unpack_info, then pack_info of the kept ray indices).  A ray's re-samples are consecutive, so: count per
// ray -> scan over RAYS -> every ray copies its kept intervals to its place and writes its own packed_info row.
__global__ __launch_bounds__(THREADS) void fg_count_kernel(int64_t n_rays, const int32_t* __restrict__ rpi, const uint8_t* __restrict__ is_fg,
                                                            int32_t* __restrict__ cnt)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int b = rpi[2 * r], s = rpi[2 * r + 1];
    int c = 0;
    for (int j = 0; j < s; j++) c += is_fg[b + j] ? 1 : 0;
    cnt[r] = c;
}

__global__ __launch_bounds__(THREADS) void fg_compact_kernel(int64_t n_rays, const int32_t* __restrict__ rpi, const uint8_t* __restrict__ is_fg,
                                                              const float* __restrict__ starts, const float* __restrict__ ends,
                                                              const int32_t* __restrict__ cnt, const int32_t* __restrict__ start,
                                                              int64_t* __restrict__ ray_indices, float* __restrict__ t_starts,
                                                              float* __restrict__ t_ends, int32_t* __restrict__ out_pinfo)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int c = cnt[r];
    const int q0 = start[r];
    out_pinfo[2 * r] = q0;                                   // (cumsum - count, count): what pack_info of the kept ray indices gives
    out_pinfo[2 * r + 1] = c;
    if (c == 0) return;
    const int b = rpi[2 * r], s = rpi[2 * r + 1];
    int q = q0;
    for (int j = 0; j < s; j++) {
        if (is_fg[b + j]) { ray_indices[q] = r; t_starts[q] = starts[b + j]; t_ends[q] = ends[b + j]; q++; }
    }
}

}  // namespace

IA_EXPORT int ia_resample_packed_info(int64_t n_rays, const int32_t* packed_info, int n, int add_steps,
                                      int32_t* resample_packed_info, int32_t* total, void* tmp, ia_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_rays == 0) return ia_exclusive_scan_i32(nullptr, nullptr, total, 0, tmp, stream);
    int32_t* cnt = (int32_t*)tmp;
    int32_t* start = cnt + n_rays;
    void* scan_tmp = (void*)(((uintptr_t)(start + n_rays) + 15) & ~(uintptr_t)15);
    const int grid = ia::cdiv(n_rays, THREADS);
    resample_counts_kernel<<<grid, THREADS, 0, s>>>(n_rays, packed_info, n, add_steps, cnt);
    int r = ia_exclusive_scan_i32(cnt, start, total, n_rays, scan_tmp, stream);
    if (r != IA_OK) return r;
    stack_info_kernel<<<grid, THREADS, 0, s>>>(n_rays, start, cnt, resample_packed_info);
    return ia::check_launch("ia_resample_packed_info");
}

IA_EXPORT int ia_pack_info(int64_t n_samples, const int64_t* ray_indices, int64_t n_rays, int32_t* packed_info,
                           void* tmp, ia_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_rays == 0) return IA_OK;
    int32_t* cnt = (int32_t*)tmp;
    int32_t* start = cnt + n_rays;
    void* scan_tmp = (void*)(((uintptr_t)(start + n_rays) + 15) & ~(uintptr_t)15);
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int32_t) * n_rays, s);
    if (e != hipSuccess) { ia::set_error("ia_pack_info: memset failed"); return IA_ERR_LAUNCH; }
    if (n_samples > 0)
        count_rays_kernel<<<ia::cdiv(n_samples, THREADS), THREADS, 0, s>>>(n_samples, ray_indices, cnt);
    int r = ia_exclusive_scan_i32(cnt, start, nullptr, n_rays, scan_tmp, stream);
    if (r != IA_OK) return r;
    stack_info_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, s>>>(n_rays, start, cnt, packed_info);
    return ia::check_launch("ia_pack_info");
}

IA_EXPORT int ia_unpack_info(int64_t n_rays, const int32_t* packed_info, int64_t* ray_indices, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    unpack_info_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, packed_info, ray_indices);
    return ia::check_launch("ia_unpack_info");
}

IA_EXPORT int ia_unpack_info_to_mask(int64_t n_rays, const int32_t* packed_info, int n_samples, uint8_t* masks,
                                     ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    unpack_mask_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, packed_info, n_samples,
                                                                                       masks);
    return ia::check_launch("ia_unpack_info_to_mask");
}

IA_EXPORT int ia_unpack_data(int64_t n_rays, const int32_t* packed_info, int data_dim, const float* data,
                             int n_samples_per_ray, float* out, ia_stream_t stream)
{
    if (n_rays == 0 || n_samples_per_ray == 0) return IA_OK;
    unpack_data_kernel<<<ia::cdiv(n_rays * n_samples_per_ray, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_rays, packed_info, data_dim, data, n_samples_per_ray, out);
    return ia::check_launch("ia_unpack_data");
}

IA_EXPORT int ia_ray_resampling(int64_t n_rays, const int32_t* packed_info, const float* starts, const float* ends,
                                const float* weights, const float* sdfs, const int32_t* resample_packed_info,
                                float* resample_ts, float* resample_offsets, int64_t* surface_idx,
                                int64_t* resample_indices, int32_t* resample_fg_counts, int32_t* resample_bg_counts,
                                ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    k1_resampling_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_rays, packed_info, starts, ends, weights, sdfs, resample_packed_info, resample_ts, resample_offsets,
        surface_idx, resample_indices, resample_fg_counts, resample_bg_counts);
    return ia::check_launch("ia_ray_resampling");
}

IA_EXPORT int ia_ray_resampling_merge(int64_t n_rays, const int32_t* packed_info, const float* vals,
                                      const uint8_t* is_left, const uint8_t* is_right, const float* weights,
                                      const int32_t* resample_packed_info, float* resample_vals,
                                      float* resample_dists, uint8_t* resample_is_left, uint8_t* resample_is_right,
                                      uint8_t* is_resample, uint8_t* is_fg_sample, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    k2_merge_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_rays, packed_info, vals, is_left, is_right, weights, resample_packed_info, resample_vals, resample_dists,
        resample_is_left, resample_is_right, is_resample, is_fg_sample);
    return ia::check_launch("ia_ray_resampling_merge");
}

IA_EXPORT int ia_ray_resampling_fine(int64_t n_rays, const int32_t* packed_info, const float* starts,
                                     const float* ends, const float* weights, const int32_t* resample_packed_info,
                                     float* resample_starts, float* resample_ends, uint8_t* is_fg_sample,
                                     ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    k34_fine_kernel<false><<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_rays, packed_info, starts, ends, weights, nullptr, resample_packed_info, resample_starts, resample_ends,
        is_fg_sample);
    return ia::check_launch("ia_ray_resampling_fine");
}

IA_EXPORT int ia_ray_resampling_sdf_fine(int64_t n_rays, const int32_t* packed_info, const float* starts,
                                         const float* ends, const float* alphas, const float* sdfs,
                                         const int32_t* resample_packed_info, float* resample_starts,
                                         float* resample_ends, uint8_t* is_fg_sample, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    k34_fine_kernel<true><<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_rays, packed_info, starts, ends, alphas, sdfs, resample_packed_info, resample_starts, resample_ends,
        is_fg_sample);
    return ia::check_launch("ia_ray_resampling_sdf_fine");
}

// count / compact the foreground intervals of a fine re-sampling; cnt, start: int32 [n_rays] (start = exclusive scan of cnt),
// total [1] = F; out_packed_info int32 [n_rays, 2] = pack_info of the kept ray indices
IA_EXPORT int ia_fg_count(int64_t n_rays, const int32_t* resampled_packed_info, const uint8_t* is_fg, int32_t* cnt, int32_t* start,
                          int32_t* total, void* scan_tmp, ia_stream_t stream)
{
    if (n_rays == 0) return ia_exclusive_scan_i32(nullptr, nullptr, total, 0, scan_tmp, stream);
    fg_count_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, resampled_packed_info, is_fg, cnt);
    int r = ia::check_launch("ia_fg_count");
    if (r != IA_OK) return r;
    return ia_exclusive_scan_i32(cnt, start, total, n_rays, scan_tmp, stream);
}

IA_EXPORT int ia_fg_compact(int64_t n_rays, const int32_t* resampled_packed_info, const uint8_t* is_fg, const float* starts, const float* ends,
                            const int32_t* cnt, const int32_t* start, int64_t* ray_indices, float* t_starts, float* t_ends,
                            int32_t* out_packed_info, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    fg_compact_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, resampled_packed_info, is_fg, starts, ends, cnt,
                                                                                      start, ray_indices, t_starts, t_ends, out_packed_info);
    return ia::check_launch("ia_fg_compact");
}
