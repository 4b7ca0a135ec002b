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
// K1..K4: only the two running sums of a ray are serial (its CDF, and the sample positions u_j, which are one table per launch);
// phase A leaves them as tables (one lane per ray), phase B inverts the CDF per OUTPUT element (binary search), so that a wave
// writes 64 consecutive elements of every output array -- see resample_math.h, which also compiles under gcc for the CPU test.
// The float expression order matches oracle/ia_oracle.c exactly (TU built with -ffp-contract=off): bit-exact outputs.
#include "ia_common.h"
#include "resample_math.h"
#include <stdlib.h>

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
// K5 as a segmented fill (the reference's pack.cu:7-28 walks a ray's samples with one thread): a wave takes 64 rays, every
// lane loads one (start, count), and the wave then fills the rays' segments one after the other, all 64 lanes storing consecutive
// elements -- a ray of the volume-interaction re-sampling (1024 entries, models/pbr/utils.py:113-135) is 16 fully coalesced 512-byte
// stores instead of 1024 scattered 8-byte ones.  Any packed_info the reference kernel accepts gives the same result (segments need not
// be ordered).
__global__ __launch_bounds__(THREADS) void unpack_info_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                               int64_t* __restrict__ ray_indices)
{
    const int lane = threadIdx.x & 63;
    const int64_t ray0 = ((int64_t)blockIdx.x * THREADS + threadIdx.x) - lane;          // first ray of this wave
    const int64_t mine = ray0 + lane;
    int2 pi = make_int2(0, 0);
    if (mine < n_rays) pi = reinterpret_cast<const int2*>(packed_info)[mine];
    unsigned long long todo = __ballot(pi.y > 0);
    while (todo) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int start = __builtin_amdgcn_readlane(pi.x, k), cnt = __builtin_amdgcn_readlane(pi.y, k);
        for (int j = lane; j < cnt; j += 64) ray_indices[(int64_t)start + j] = ray0 + k;
    }
}

// one lane per (ray, slot) of the [n_rays, n_samples] mask
__global__ __launch_bounds__(THREADS) void unpack_mask_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                               int n_samples, uint8_t* __restrict__ masks)
{
    const int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (t >= n_rays * (int64_t)n_samples) return;
    const int64_t r = t / n_samples;
    const int j = (int)(t - r * n_samples);
    if (j < packed_info[2 * r + 1]) masks[t] = 1;
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

// ---- K1 .. K4: per-ray CDF tables + per-element inversion (arithmetic: resample_math.h) -----------------------------------------
// Scratch of one call (ia_resample_tmp_bytes): the u-table, cdf / cmax per input interval, one 16-byte record per ray, the
// rank -> ray list of the non-empty rays (every non-empty ray owns exactly n consecutive outputs, K1 / K3 / K4), and for K2 the
// first[] ranks per edge.
struct RsScratch {
    float* utab;
    float* cdf;
    float* cmax;
    ia_rs_ray* ray;
    int32_t* rank2ray;
    int32_t* first;
};

constexpr size_t rs_align(size_t v) { return (v + 255) & ~(size_t)255; }

size_t rs_layout(void* tmp, int64_t n_rays, int64_t n_in, int n, RsScratch* s)
{
    char* b = reinterpret_cast<char*>(tmp);
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = b ? b + o : nullptr; o += rs_align(bytes); return p; };
    float* utab = reinterpret_cast<float*>(take(sizeof(float) * (size_t)(n + 2)));
    float* cdf = reinterpret_cast<float*>(take(sizeof(float) * (size_t)(n_in + 1)));
    float* cmax = reinterpret_cast<float*>(take(sizeof(float) * (size_t)(n_in + 1)));
    ia_rs_ray* ray = reinterpret_cast<ia_rs_ray*>(take(sizeof(ia_rs_ray) * (size_t)(n_rays + 1)));
    int32_t* rank2ray = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * (size_t)(n_rays + 1)));
    int32_t* first = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * (size_t)(n_in + 1)));
    if (s) { s->utab = utab; s->cdf = cdf; s->cmax = cmax; s->ray = ray; s->rank2ray = rank2ray; s->first = first; }
    return o;
}

// the launch's sample positions: n serial fp32 additions, once (every ray of the reference repeats them)
__global__ void rs_utab_kernel(int n, int fine, float* __restrict__ utab)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) ia_rs_fill_utab(n, fine, utab);
}

// K1 phase A: one lane per ray
__global__ __launch_bounds__(THREADS) void rs1_rays_kernel(int64_t n_rays, int n, const int32_t* __restrict__ packed_info,
                                                           const int32_t* __restrict__ rpi, const float* __restrict__ starts,
                                                           const float* __restrict__ ends, const float* __restrict__ weights,
                                                           const float* __restrict__ sdfs, RsScratch s, int64_t* __restrict__ surface_idx,
                                                           int32_t* __restrict__ fg_counts, int32_t* __restrict__ bg_counts)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    if (pi.y == 0) { surface_idx[r] = -1; bg_counts[r] = 0; return; }
    ia_rs_ray rec;
    int32_t surf, bg;
    ia_rs1_ray(pi.y, weights + pi.x, sdfs + pi.x, starts + pi.x, ends + pi.x, s.utab, n, s.cdf + pi.x, s.cmax + pi.x, fg_counts + pi.x, &rec,
               &surf, &bg);
    surface_idx[r] = surf >= 0 ? (int64_t)surf + pi.x : -1;
    bg_counts[r] = bg;
    reinterpret_cast<int4*>(s.ray)[r] = make_int4(rec.n_hit, rec.j_clamp, __float_as_int(rec.v_clamp), rec.k_first);
    s.rank2ray[rpi[2 * r] / n] = (int32_t)r;
}

// K1 phase B: one lane per re-sample; a wave writes 64 consecutive elements of ts / offsets / indices
__global__ __launch_bounds__(THREADS) void rs1_samples_kernel(int64_t n_out, const int32_t* __restrict__ n_out_dev, int n,
                                                              const int32_t* __restrict__ packed_info,
                                                              const float* __restrict__ starts, const float* __restrict__ ends, RsScratch s,
                                                              float* __restrict__ ts, float* __restrict__ offsets, int64_t* __restrict__ indices)
{
    const int64_t e = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (e >= n_out) return;
    if (n_out_dev && e >= (int64_t)*n_out_dev) return;      // capacity-sized launch: the total is still on the device (ranks behind it own no ray)
    const uint32_t rank = (uint32_t)e / (uint32_t)n;
    const int j = (int)((uint32_t)e - rank * (uint32_t)n);
    const int r = s.rank2ray[rank];
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    const int4 q = reinterpret_cast<const int4*>(s.ray)[r];
    ia_rs_ray rec;
    rec.n_hit = q.x; rec.j_clamp = q.y; rec.v_clamp = __int_as_float(q.z); rec.k_first = q.w;
    float t, off;
    int32_t k;
    ia_rs1_sample(j, pi.y, &rec, starts + pi.x, ends + pi.x, s.cdf + pi.x, s.cmax + pi.x, s.utab, &t, &off, &k);
    ts[e] = t;
    offsets[e] = off;
    indices[e] = (int64_t)k + pi.x;
}

// K2 phase A: one lane per ray (edge list)
__global__ __launch_bounds__(THREADS) void rs2_rays_kernel(int64_t n_rays, int n, const int32_t* __restrict__ packed_info,
                                                           const float* __restrict__ vals, const uint8_t* __restrict__ il,
                                                           const uint8_t* __restrict__ ir, const float* __restrict__ weights, RsScratch s)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    if (pi.y == 0) return;
    ia_rs_ray rec;
    ia_rs2_ray(pi.y, vals + pi.x, il + pi.x, ir + pi.x, weights + pi.x, s.utab, n, s.cdf + pi.x, s.cmax + pi.x, s.first + pi.x, &rec);
    reinterpret_cast<int4*>(s.ray)[r] = make_int4(rec.n_hit, rec.j_clamp, __float_as_int(rec.v_clamp), rec.k_first);
}

// K2 phase B: a workgroup owns RS2_TILE consecutive rays = one contiguous range of the merged edge list; one lane per output edge
// (ray by binary search in the tile's LDS offsets), every output array written in element order, unreached slots as zeros
constexpr int RS2_TILE = 64;
__global__ __launch_bounds__(THREADS) void rs2_edges_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                            const int32_t* __restrict__ rpi, const float* __restrict__ vals,
                                                            const uint8_t* __restrict__ il, const uint8_t* __restrict__ ir, RsScratch s,
                                                            float* __restrict__ out_vals, float* __restrict__ out_dists,
                                                            uint8_t* __restrict__ out_left, uint8_t* __restrict__ out_right,
                                                            uint8_t* __restrict__ out_resample, uint8_t* __restrict__ out_fg)
{
    __shared__ int s_off[RS2_TILE + 1];
    const int64_t r0 = (int64_t)blockIdx.x * RS2_TILE;
    const int n_tile = (int)((r0 + RS2_TILE <= n_rays) ? RS2_TILE : n_rays - r0);
    if (threadIdx.x < n_tile) s_off[threadIdx.x] = rpi[2 * (r0 + threadIdx.x)];
    if (threadIdx.x == 0) s_off[n_tile] = rpi[2 * (r0 + n_tile - 1)] + rpi[2 * (r0 + n_tile - 1) + 1];
    __syncthreads();
    const int begin = s_off[0], end = s_off[n_tile];
    for (int p = begin + threadIdx.x; p < end; p += THREADS) {
        int lo = 0, hi = n_tile - 1;                         // last ray of the tile whose range starts at or before p
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= p) lo = mid; else hi = mid - 1;
        }
        const int64_t r = r0 + lo;
        const int local = p - s_off[lo], cnt = s_off[lo + 1] - s_off[lo];
        const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
        const int4 q = reinterpret_cast<const int4*>(s.ray)[r];
        ia_rs_ray rec;
        rec.n_hit = q.x; rec.j_clamp = q.y; rec.v_clamp = __int_as_float(q.z); rec.k_first = q.w;
        const ia_rs2_edge e = ia_rs2_at(local, pi.y, &rec, vals + pi.x, il + pi.x, ir + pi.x, s.cdf + pi.x, s.first + pi.x, s.utab);
        float d = 0.0f;
        if (local + 1 < cnt) {
            const ia_rs2_edge f = ia_rs2_at(local + 1, pi.y, &rec, vals + pi.x, il + pi.x, ir + pi.x, s.cdf + pi.x, s.first + pi.x, s.utab);
            if (f.used) d = f.val - e.val;
        }
        out_vals[p] = e.val;
        out_dists[p] = d;
        out_left[p] = e.left;
        out_right[p] = e.right;
        out_resample[p] = e.resample;
        out_fg[p] = e.used;
    }
}

// K2 with the caller's compaction fused in (models/intrinsic_avatar.py:1221-1226 keeps the edges with is_fg and re-packs their ray
// indices: nonzero + four boolean-mask gathers + unpack_info + pack_info).  A ray's reached edges are the first steps + n_hit slots of
// its range, so: per-ray count -> scan over RAYS -> phase B writes every kept edge straight to its final place, with its ray index,
// and the ray's packed_info row.  Nothing is zero-filled, nothing is written twice.
__global__ __launch_bounds__(THREADS) void rs2_counts_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info, RsScratch s,
                                                             int32_t* __restrict__ cnt)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int steps = packed_info[2 * r + 1];
    cnt[r] = steps > 0 ? steps + reinterpret_cast<const int4*>(s.ray)[r].x : 0;
}

__global__ __launch_bounds__(THREADS) void rs2_compact_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                              const int32_t* __restrict__ cnt, const int32_t* __restrict__ start,
                                                              const float* __restrict__ vals, const uint8_t* __restrict__ il,
                                                              const uint8_t* __restrict__ ir, RsScratch s, float* __restrict__ out_vals,
                                                              uint8_t* __restrict__ out_left, uint8_t* __restrict__ out_right,
                                                              int64_t* __restrict__ out_ray, int32_t* __restrict__ out_pinfo)
{
    __shared__ int s_off[RS2_TILE + 1];
    const int64_t r0 = (int64_t)blockIdx.x * RS2_TILE;
    const int n_tile = (int)((r0 + RS2_TILE <= n_rays) ? RS2_TILE : n_rays - r0);
    if (threadIdx.x < n_tile) {
        const int st = start[r0 + threadIdx.x], c = cnt[r0 + threadIdx.x];
        s_off[threadIdx.x] = st;
        reinterpret_cast<int2*>(out_pinfo)[r0 + threadIdx.x] = make_int2(st, c);
        if (threadIdx.x == n_tile - 1) s_off[n_tile] = st + c;
    }
    __syncthreads();
    const int begin = s_off[0], end = s_off[n_tile];
    for (int p = begin + threadIdx.x; p < end; p += THREADS) {
        int lo = 0, hi = n_tile - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= p) lo = mid; else hi = mid - 1;
        }
        const int64_t r = r0 + lo;
        const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
        const int4 q = reinterpret_cast<const int4*>(s.ray)[r];
        ia_rs_ray rec;
        rec.n_hit = q.x; rec.j_clamp = q.y; rec.v_clamp = __int_as_float(q.z); rec.k_first = q.w;
        const ia_rs2_edge e = ia_rs2_at(p - s_off[lo], pi.y, &rec, vals + pi.x, il + pi.x, ir + pi.x, s.cdf + pi.x, s.first + pi.x, s.utab);
        out_vals[p] = e.val;
        out_left[p] = e.left;
        out_right[p] = e.right;
        out_ray[p] = r;
    }
}

// K3 / K4, few points per ray (n + 1 <= IA_RS_SMALL): one lane per ray, the ray's points in registers, its n outputs written as one
// vector per array (the non-empty rays' outputs are consecutive, so a wave's stores are contiguous)
template <bool SDF, int N>
__global__ __launch_bounds__(THREADS) void rs34_small_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                             const int32_t* __restrict__ rpi, const float* __restrict__ starts,
                                                             const float* __restrict__ ends, const float* __restrict__ wa,
                                                             const float* __restrict__ sdfs, float du, float u0,
                                                             float* __restrict__ out_starts, float* __restrict__ out_ends,
                                                             uint8_t* __restrict__ out_fg)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    if (pi.y == 0) return;
    float pts[IA_RS_SMALL];
#pragma unroll
    for (int j = 0; j < IA_RS_SMALL; j++) pts[j] = 0.0f;
    const int hit = ia_rs34_small(SDF ? 1 : 0, N + 1, pi.y, wa + pi.x, SDF ? sdfs + pi.x : nullptr, starts + pi.x, ends + pi.x, du, u0, pts);
    const int rb = rpi[2 * r];
    float os[N], oe[N];
    uint8_t fg[N];
#pragma unroll
    for (int q = 0; q < N; q++) {
        os[q] = q < hit ? pts[q] : 0.0f;
        oe[q] = q + 1 < hit ? pts[q + 1] : 0.0f;
        fg[q] = (uint8_t)(q + 1 < hit);
    }
    if (N == 4) {
        *reinterpret_cast<float4*>(out_starts + rb) = make_float4(os[0], os[1], os[2], os[3]);
        *reinterpret_cast<float4*>(out_ends + rb) = make_float4(oe[0], oe[1], oe[2], oe[3]);
        *reinterpret_cast<uchar4*>(out_fg + rb) = make_uchar4(fg[0], fg[1], fg[2], fg[3]);
    } else {
#pragma unroll
        for (int q = 0; q < N; q++) { out_starts[rb + q] = os[q]; out_ends[rb + q] = oe[q]; out_fg[rb + q] = fg[q]; }
    }
}

// K3 / K4, general n: phase A (tables) + phase B (one lane per output interval)
template <bool SDF>
__global__ __launch_bounds__(THREADS) void rs34_rays_kernel(int64_t n_rays, int n, const int32_t* __restrict__ packed_info,
                                                            const int32_t* __restrict__ rpi, const float* __restrict__ wa,
                                                            const float* __restrict__ sdfs, RsScratch s)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    if (pi.y == 0) return;
    ia_rs_ray rec;
    ia_rs34_ray(SDF ? 1 : 0, pi.y, wa + pi.x, SDF ? sdfs + pi.x : nullptr, s.utab, n + 1, s.cdf + pi.x, s.cmax + pi.x, &rec);
    reinterpret_cast<int4*>(s.ray)[r] = make_int4(rec.n_hit, rec.j_clamp, __float_as_int(rec.v_clamp), rec.k_first);
    s.rank2ray[rpi[2 * r] / n] = (int32_t)r;
}

__global__ __launch_bounds__(THREADS) void rs34_intervals_kernel(int64_t n_out, int n, const int32_t* __restrict__ packed_info,
                                                                 const float* __restrict__ starts, const float* __restrict__ ends, RsScratch s,
                                                                 float* __restrict__ out_starts, float* __restrict__ out_ends,
                                                                 uint8_t* __restrict__ out_fg)
{
    const int64_t e = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (e >= n_out) return;
    const uint32_t rank = (uint32_t)e / (uint32_t)n;
    const int q = (int)((uint32_t)e - rank * (uint32_t)n);
    const int r = s.rank2ray[rank];
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    const int4 h = reinterpret_cast<const int4*>(s.ray)[r];
    ia_rs_ray rec;
    rec.n_hit = h.x; rec.j_clamp = h.y; rec.v_clamp = __int_as_float(h.z); rec.k_first = h.w;
    const bool has_s = q < rec.n_hit, has_e = q + 1 < rec.n_hit;
    out_starts[e] = has_s ? ia_rs34_point(q, pi.y, &rec, starts + pi.x, ends + pi.x, s.cdf + pi.x, s.cmax + pi.x, s.utab) : 0.0f;
    out_ends[e] = has_e ? ia_rs34_point(q + 1, pi.y, &rec, starts + pi.x, ends + pi.x, s.cdf + pi.x, s.cmax + pi.x, s.utab) : 0.0f;
    out_fg[e] = (uint8_t)has_e;
}


// ---- foreground compaction of a fine re-sampling (K3 / K4 outputs) -----------------------------------------------------------
// The caller of ray_resampling_sdf_fine keeps the foreground intervals only (models/intrinsic_avatar.py:516-528: three
// boolean-mask gathers + unpack_info, then pack_info of the kept ray indices).  A ray's re-samples are consecutive, so: count per
// ray -> scan over RAYS -> every ray copies its kept intervals to its place and writes its own packed_info row.
// ---- samples of an interval (edge) list: element-parallel over EDGES (a scan of the left-edge flags places every sample; every
// output is written by consecutive lanes).  pos[e] = number of left edges before e.
__global__ __launch_bounds__(THREADS) void left_flags_kernel(int64_t n_edges, const uint8_t* __restrict__ is_left, int32_t* __restrict__ flag)
{
    const int64_t e = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (e < n_edges) flag[e] = is_left[e] ? 1 : 0;
}

// the same over a CAPACITY-sized edge list whose length is still on the device (*n_edges <= capacity): slots behind the list count nothing
__global__ __launch_bounds__(THREADS) void left_flags_upto_kernel(int64_t capacity, const int32_t* __restrict__ n_edges,
                                                                   const uint8_t* __restrict__ is_left, int32_t* __restrict__ flag)
{
    const int64_t e = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (e < capacity) flag[e] = (e < (int64_t)*n_edges && is_left[e]) ? 1 : 0;
}

__global__ __launch_bounds__(THREADS) void interval_samples_kernel(int64_t n_rays, int64_t n_edges, const int32_t* __restrict__ edge_pinfo,
                                                                   const float* __restrict__ vals, const int64_t* __restrict__ ray_indices,
                                                                   const uint8_t* __restrict__ is_left, const int32_t* __restrict__ pos,
                                                                   const int32_t* __restrict__ total, int64_t* __restrict__ left_idx,
                                                                   float* __restrict__ t_starts, float* __restrict__ t_ends,
                                                                   int64_t* __restrict__ out_ray, int32_t* __restrict__ out_pinfo)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i < n_edges && is_left[i]) {
        const int32_t p = pos[i];
        if (left_idx) left_idx[p] = i;
        t_starts[p] = vals[i];
        t_ends[p] = vals[i + 1 < n_edges ? i + 1 : i];                 // a left edge is followed by its right edge
        out_ray[p] = ray_indices[i];
    }
    if (i < n_rays) {                                                  // pack_info of the samples' ray indices
        // (start, count) as lib/nerfacc/pack.py:72-75 forms them: start = samples in front of the ray, also for a ray without samples
        const int32_t s0 = edge_pinfo[2 * i], c = edge_pinfo[2 * i + 1];
        const int32_t a = (s0 < n_edges) ? pos[s0] : *total;
        const int32_t b = ((int64_t)s0 + c < n_edges) ? pos[s0 + c] : *total;
        out_pinfo[2 * i] = a;
        out_pinfo[2 * i + 1] = b - a;
    }
}

__global__ __launch_bounds__(THREADS) void samples_to_edges_kernel(int64_t n_edges, const uint8_t* __restrict__ is_left,
                                                                   const int32_t* __restrict__ pos, const float* __restrict__ sample_vals,
                                                                   float fill, float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (e < n_edges) out[e] = is_left[e] ? sample_vals[pos[e]] : fill;
}

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
    if (n_rays == 0 || n_samples <= 0) return IA_OK;
    unpack_mask_kernel<<<ia::cdiv(n_rays * (int64_t)n_samples, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, packed_info, n_samples,
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

IA_EXPORT size_t ia_resample_tmp_bytes(int64_t n_rays, int64_t n_in, int n) { return rs_layout(nullptr, n_rays, n_in, n, nullptr) + 256; }

static void* rs_aligned(void* tmp) { return (void*)(((uintptr_t)tmp + 255) & ~(uintptr_t)255); }

IA_EXPORT int ia_ray_resampling(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts, const float* ends,
                                const float* weights, const float* sdfs, const int32_t* resample_packed_info, int64_t n_out,
                                float* resample_ts, float* resample_offsets, int64_t* surface_idx,
                                int64_t* resample_indices, int32_t* resample_fg_counts, int32_t* resample_bg_counts, void* tmp,
                                ia_stream_t stream)
{
    return ia_ray_resampling_upto(n_rays, n_in, n, packed_info, starts, ends, weights, sdfs, resample_packed_info, n_out, nullptr, resample_ts,
                                  resample_offsets, surface_idx, resample_indices, resample_fg_counts, resample_bg_counts, tmp, stream);
}

// the same with the outputs sized for n_out = n x n_rays slots (every ray hit) and the true total -- ia_resample_packed_info's, n x (rays
// with samples) -- still on the device: the per-output phase stops at *n_out_dev, the slots behind it stay unwritten
IA_EXPORT int ia_ray_resampling_upto(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts, const float* ends,
                                     const float* weights, const float* sdfs, const int32_t* resample_packed_info, int64_t n_out,
                                     const int32_t* n_out_dev, float* resample_ts, float* resample_offsets, int64_t* surface_idx,
                                     int64_t* resample_indices, int32_t* resample_fg_counts, int32_t* resample_bg_counts, void* tmp,
                                     ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(n >= 2, "ia_ray_resampling: n must be >= 2 (cdf.py:49)");
    IA_REQUIRE(n_out >= 0 && n_out < ((int64_t)1 << 31) && n_out % n == 0, "ia_ray_resampling: n_out must be n x (rays with samples), below 2^31");
    IA_REQUIRE(tmp != nullptr, "ia_ray_resampling: tmp (ia_resample_tmp_bytes) is required");
    hipStream_t st = (hipStream_t)stream;
    RsScratch s;
    rs_layout(rs_aligned(tmp), n_rays, n_in, n, &s);
    rs_utab_kernel<<<1, 64, 0, st>>>(n, 0, s.utab);
    rs1_rays_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, st>>>(n_rays, n, packed_info, resample_packed_info, starts, ends, weights, sdfs, s,
                                                                   surface_idx, resample_fg_counts, resample_bg_counts);
    if (n_out > 0)
        rs1_samples_kernel<<<ia::cdiv(n_out, THREADS), THREADS, 0, st>>>(n_out, n_out_dev, n, packed_info, starts, ends, s, resample_ts,
                                                                         resample_offsets, resample_indices);
    return ia::check_launch("ia_ray_resampling");
}

IA_EXPORT int ia_ray_resampling_merge(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* vals,
                                      const uint8_t* is_left, const uint8_t* is_right, const float* weights,
                                      const int32_t* resample_packed_info, float* resample_vals,
                                      float* resample_dists, uint8_t* resample_is_left, uint8_t* resample_is_right,
                                      uint8_t* is_resample, uint8_t* is_fg_sample, void* tmp, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(n >= 1, "ia_ray_resampling_merge: n must be >= 1");
    IA_REQUIRE(tmp != nullptr, "ia_ray_resampling_merge: tmp (ia_resample_tmp_bytes) is required");
    hipStream_t st = (hipStream_t)stream;
    RsScratch s;
    rs_layout(rs_aligned(tmp), n_rays, n_in, n, &s);
    rs_utab_kernel<<<1, 64, 0, st>>>(n, 0, s.utab);
    rs2_rays_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, st>>>(n_rays, n, packed_info, vals, is_left, is_right, weights, s);
    rs2_edges_kernel<<<ia::cdiv(n_rays, RS2_TILE), THREADS, 0, st>>>(n_rays, packed_info, resample_packed_info, vals, is_left, is_right, s,
                                                                     resample_vals, resample_dists, resample_is_left, resample_is_right,
                                                                     is_resample, is_fg_sample);
    return ia::check_launch("ia_ray_resampling_merge");
}

// K2 + the caller's foreground compaction in two calls around ONE size read-back:
//   ia_ray_resampling_merge_count: u-table, per-ray tables, cnt [n_rays] = kept edges per ray, start = exclusive scan, *total;
//   ia_ray_resampling_merge_fill:  vals / is_left / is_right / ray_indices [total] and packed_info [n_rays, 2] of the kept edges.
// tmp (ia_resample_tmp_bytes) carries the tables from the first call to the second.
IA_EXPORT int ia_ray_resampling_merge_count(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* vals,
                                            const uint8_t* is_left, const uint8_t* is_right, const float* weights, int32_t* cnt,
                                            int32_t* start, int32_t* total, void* tmp, void* scan_tmp, ia_stream_t stream)
{
    if (n_rays == 0) return ia_exclusive_scan_i32(nullptr, nullptr, total, 0, scan_tmp, stream);
    IA_REQUIRE(n >= 1, "ia_ray_resampling_merge_count: n must be >= 1");
    IA_REQUIRE(tmp != nullptr, "ia_ray_resampling_merge_count: tmp (ia_resample_tmp_bytes) is required");
    hipStream_t st = (hipStream_t)stream;
    RsScratch s;
    rs_layout(rs_aligned(tmp), n_rays, n_in, n, &s);
    const int grid = ia::cdiv(n_rays, THREADS);
    rs_utab_kernel<<<1, 64, 0, st>>>(n, 0, s.utab);
    rs2_rays_kernel<<<grid, THREADS, 0, st>>>(n_rays, n, packed_info, vals, is_left, is_right, weights, s);
    rs2_counts_kernel<<<grid, THREADS, 0, st>>>(n_rays, packed_info, s, cnt);
    int r = ia::check_launch("ia_ray_resampling_merge_count");
    if (r != IA_OK) return r;
    return ia_exclusive_scan_i32(cnt, start, total, n_rays, scan_tmp, stream);
}

IA_EXPORT int ia_ray_resampling_merge_fill(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* vals,
                                           const uint8_t* is_left, const uint8_t* is_right, const int32_t* cnt, const int32_t* start,
                                           float* out_vals, uint8_t* out_is_left, uint8_t* out_is_right, int64_t* out_ray_indices,
                                           int32_t* out_packed_info, void* tmp, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(tmp != nullptr, "ia_ray_resampling_merge_fill: tmp of the matching ia_ray_resampling_merge_count call is required");
    RsScratch s;
    rs_layout(rs_aligned(tmp), n_rays, n_in, n, &s);
    rs2_compact_kernel<<<ia::cdiv(n_rays, RS2_TILE), THREADS, 0, (hipStream_t)stream>>>(n_rays, packed_info, cnt, start, vals, is_left, is_right,
                                                                                       s, out_vals, out_is_left, out_is_right,
                                                                                       out_ray_indices, out_packed_info);
    return ia::check_launch("ia_ray_resampling_merge_fill");
}

template <bool SDF>
static int launch_fine(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts, const float* ends,
                       const float* wa, const float* sdfs, const int32_t* rpi, int64_t n_out, float* out_starts, float* out_ends,
                       uint8_t* out_fg, void* tmp, hipStream_t st, const char* what)
{
    if (n_rays == 0) return IA_OK;
    const int grid = ia::cdiv(n_rays, THREADS);
    if (n + 1 <= IA_RS_SMALL && getenv("IA_RESAMPLE_TABLES") == nullptr) {
        const int bins = n + 1;
        const float du = (float)((1.0f - 1.0 / bins) / n);
        const float u0 = (float)(1.0 / (2 * bins));
#define IA_RS_CASE(N)                                                                                                                  \
    case N:                                                                                                                            \
        rs34_small_kernel<SDF, N><<<grid, THREADS, 0, st>>>(n_rays, packed_info, rpi, starts, ends, wa, sdfs, du, u0, out_starts, out_ends, \
                                                            out_fg);                                                                   \
        break;
        switch (n) { IA_RS_CASE(1) IA_RS_CASE(2) IA_RS_CASE(3) IA_RS_CASE(4) IA_RS_CASE(5) IA_RS_CASE(6) IA_RS_CASE(7) IA_RS_CASE(8) }
#undef IA_RS_CASE
        return ia::check_launch(what);
    }
    if (tmp == nullptr) { ia::set_error("%s: tmp (ia_resample_tmp_bytes) is required", what); return IA_ERR_INVALID; }
    if (!(n_out >= 0 && n_out < ((int64_t)1 << 31) && n_out % n == 0)) { ia::set_error("%s: n_out must be n x (rays with samples), below 2^31", what); return IA_ERR_INVALID; }
    RsScratch s;
    rs_layout(rs_aligned(tmp), n_rays, n_in, n, &s);
    rs_utab_kernel<<<1, 64, 0, st>>>(n, 1, s.utab);
    rs34_rays_kernel<SDF><<<grid, THREADS, 0, st>>>(n_rays, n, packed_info, rpi, wa, sdfs, s);
    if (n_out > 0)
        rs34_intervals_kernel<<<ia::cdiv(n_out, THREADS), THREADS, 0, st>>>(n_out, n, packed_info, starts, ends, s, out_starts, out_ends, out_fg);
    return ia::check_launch(what);
}

IA_EXPORT int ia_ray_resampling_fine(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts,
                                     const float* ends, const float* weights, const int32_t* resample_packed_info, int64_t n_out,
                                     float* resample_starts, float* resample_ends, uint8_t* is_fg_sample, void* tmp,
                                     ia_stream_t stream)
{
    IA_REQUIRE(n >= 1, "ia_ray_resampling_fine: n must be >= 1");
    return launch_fine<false>(n_rays, n_in, n, packed_info, starts, ends, weights, nullptr, resample_packed_info, n_out, resample_starts,
                              resample_ends, is_fg_sample, tmp, (hipStream_t)stream, "ia_ray_resampling_fine");
}

IA_EXPORT int ia_ray_resampling_sdf_fine(int64_t n_rays, int64_t n_in, int n, const int32_t* packed_info, const float* starts,
                                         const float* ends, const float* alphas, const float* sdfs,
                                         const int32_t* resample_packed_info, int64_t n_out, float* resample_starts,
                                         float* resample_ends, uint8_t* is_fg_sample, void* tmp, ia_stream_t stream)
{
    IA_REQUIRE(n >= 1, "ia_ray_resampling_sdf_fine: n must be >= 1");
    return launch_fine<true>(n_rays, n_in, n, packed_info, starts, ends, alphas, sdfs, resample_packed_info, n_out, resample_starts,
                             resample_ends, is_fg_sample, tmp, (hipStream_t)stream, "ia_ray_resampling_sdf_fine");
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

// the samples of an interval list (models/intrinsic_avatar.py:1242-1247 and :1000-1030: vals[is_left], vals[is_right],
// ray_indices[is_left], pack_info) as flag -> scan -> fill over the EDGES: pos int32 [n_edges] (exclusive scan of is_left), *total = S
IA_EXPORT int ia_interval_samples_count(int64_t n_edges, const uint8_t* is_left, int32_t* pos, int32_t* total, void* scan_tmp,
                                        ia_stream_t stream)
{
    if (n_edges == 0) return ia_exclusive_scan_i32(nullptr, nullptr, total, 0, scan_tmp, stream);
    IA_REQUIRE(n_edges < ((int64_t)1 << 31), "ia_interval_samples_count: n_edges must be below 2^31");
    left_flags_kernel<<<ia::cdiv(n_edges, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_edges, is_left, pos);
    int r = ia::check_launch("ia_interval_samples_count");
    if (r != IA_OK) return r;
    return ia_exclusive_scan_i32(pos, pos, total, n_edges, scan_tmp, stream);
}

IA_EXPORT int ia_interval_samples_count_upto(int64_t capacity, const uint8_t* is_left, const int32_t* n_edges, int32_t* pos, int32_t* total,
                                             void* scan_tmp, ia_stream_t stream)
{
    if (capacity == 0) return ia_exclusive_scan_i32(nullptr, nullptr, total, 0, scan_tmp, stream);
    IA_REQUIRE(capacity < ((int64_t)1 << 31), "ia_interval_samples_count_upto: capacity must be below 2^31");
    IA_REQUIRE(n_edges != nullptr, "ia_interval_samples_count_upto: the device-side length is required");
    left_flags_upto_kernel<<<ia::cdiv(capacity, THREADS), THREADS, 0, (hipStream_t)stream>>>(capacity, n_edges, is_left, pos);
    int r = ia::check_launch("ia_interval_samples_count_upto");
    if (r != IA_OK) return r;
    return ia_exclusive_scan_i32(pos, pos, total, capacity, scan_tmp, stream);
}

IA_EXPORT int ia_interval_samples_fill(int64_t n_rays, int64_t n_edges, const int32_t* edge_packed_info, const float* vals,
                                       const int64_t* ray_indices, const uint8_t* is_left, const int32_t* pos, const int32_t* total,
                                       int64_t* left_idx, float* t_starts, float* t_ends, int64_t* sample_ray_indices,
                                       int32_t* sample_packed_info, ia_stream_t stream)
{
    const int64_t n = n_rays > n_edges ? n_rays : n_edges;
    if (n == 0) return IA_OK;
    interval_samples_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, n_edges, edge_packed_info, vals, ray_indices,
                                                                                      is_left, pos, total, left_idx, t_starts, t_ends,
                                                                                      sample_ray_indices, sample_packed_info);
    return ia::check_launch("ia_interval_samples_fill");
}

// out[e] = is_left[e] ? sample_vals[pos[e]] : fill -- a per-sample quantity back on the edge list (the scatter of alpha_fn's
// `sdf[is_left] = ...`, models/intrinsic_avatar.py:1017-1027, as a gather with coalesced stores)
IA_EXPORT int ia_samples_to_edges(int64_t n_edges, const uint8_t* is_left, const int32_t* pos, const float* sample_vals, float fill,
                                  float* out, ia_stream_t stream)
{
    if (n_edges == 0) return IA_OK;
    samples_to_edges_kernel<<<ia::cdiv(n_edges, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_edges, is_left, pos, sample_vals, fill, out);
    return ia::check_launch("ia_samples_to_edges");
}
