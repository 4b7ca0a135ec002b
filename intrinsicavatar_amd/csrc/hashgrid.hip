// hashgrid.hip -- multiresolution hash-grid encoding (+ spherical harmonics deg 4) for gfx950.
// Replaces tinycudann.Encoding(HashGrid) / Encoding(SphericalHarmonics) as used by the reference
// (models/network_utils.py:58-100,191; configs/geometry/progressive_hash_grid.yaml:9-24;
//  configs/radiance/progressive_hash_grid.yaml:5-19).  tiny-cuda-nn is not vendored in the
// reference tree; semantics follow oracle/ia_oracle_field.c (Instant-NGP definitions).
//
// Mapping: one lane per (point, level), level fastest.  A wave covers 4 points x 16 levels, so
//   * the 8 corner gathers of 64 lanes (512 independent 8-byte loads) are all in flight at once --
//     the kernel is gather (L2 / Infinity-Cache) bound, never ALU bound;
//   * each point's 32 outputs (16 levels x 2 features) are written by 16 adjacent lanes as one
//     contiguous 128-byte row: fully coalesced stores;
//   * per-level constants (scale, resolution, table offset/size) live in registers, computed once.
// Tables are fp32 [entries, 2] (50.4 MB per grid): resident in the 256 MiB Infinity Cache.
#include <stdlib.h>

#include "ia_common.h"

namespace {

constexpr int THREADS = 256;
constexpr int MAX_LEVELS = 32;

struct HashCfg {
    int n_levels;
    uint32_t offsets[MAX_LEVELS + 1];
    uint32_t res[MAX_LEVELS];
    float scale[MAX_LEVELS];
};

__host__ void make_cfg(HashCfg& c, int n_levels, int log2_hashmap_size, int base_resolution, float per_level_scale)
{
    c.n_levels = n_levels;
    uint32_t offset = 0;
    const float l2 = log2f(per_level_scale);
    for (int l = 0; l < n_levels; l++) {
        const float sc = exp2f((float)l * l2) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(sc) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2;
        uint32_t p = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
        p = (p + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (p > cap) p = cap;
        c.offsets[l] = offset;
        c.res[l] = res;
        c.scale[l] = sc;
        offset += p;
    }
    c.offsets[n_levels] = offset;
}

__device__ __forceinline__ uint32_t grid_index(uint32_t hsize, uint32_t res, uint32_t px, uint32_t py, uint32_t pz)
{
    // tiny-cuda-nn grid_index<3>: dense while the stride fits, else coherent prime hash
    uint32_t stride = 1, index = 0;
    bool hashed = false;
    index += px * stride; stride *= res;
    if (stride <= hsize) { index += py * stride; stride *= res; } else hashed = true;
    if (!hashed && stride <= hsize) { index += pz * stride; stride *= res; } else hashed = true;
    if (hsize < stride) index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
    // hashed levels have a power-of-two table (2^log2_hashmap_size): mask instead of the ~25-instruction 32-bit modulo
    // (8 corners x 16 levels per point); identical result
    return ((hsize & (hsize - 1u)) == 0u) ? (index & (hsize - 1u)) : (index % hsize);
}

typedef float f4_a8 __attribute__((ext_vector_type(4), aligned(8)));

// forward (+ optional analytic d enc / d x)
template <bool WITH_JAC>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_fwd_kernel(int64_t n, const float* __restrict__ x,
                                                            const float2* __restrict__ params, HashCfg cfg,
                                                            float* __restrict__ out, int out_stride,
                                                            float* __restrict__ dy_dx)
{
    const int L = cfg.n_levels;
    const int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t i = t / L;
    const int l = (int)(t % L);
    if (i >= n) return;
    const float sc = cfg.scale[l];
    const uint32_t res = cfg.res[l], hsize = cfg.offsets[l + 1] - cfg.offsets[l];
    const float2* tab = params + cfg.offsets[l];
    float pos[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float p = fmaf(sc, x[i * 3 + d], 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)(int)fl;
        pos[d] = p - fl;
    }
    float2 v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        // plain 8-byte gathers here: this kernel is bound by the sector traffic through the fabric, not by request count
        // (pairing the x-neighbours, as xcd_gather does, measured 2.79 vs 2.72 ms)
        const uint32_t idx = grid_index(hsize, res, pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1));
        v[c] = tab[idx];
    }
    float a0 = 0.f, a1 = 0.f;
    float j0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float wx = (c & 1) ? pos[0] : 1.0f - pos[0];
        const float wy = (c & 2) ? pos[1] : 1.0f - pos[1];
        const float wz = (c & 4) ? pos[2] : 1.0f - pos[2];
        const float w = wx * wy * wz;
        a0 = fmaf(w, v[c].x, a0);       // explicit: the flat and the level-major kernels must round alike whatever hipcc contracts
        a1 = fmaf(w, v[c].y, a1);
        if (WITH_JAC) {
            const float dx = ((c & 1) ? sc : -sc) * wy * wz;
            const float dy = ((c & 2) ? sc : -sc) * wx * wz;
            const float dz = ((c & 4) ? sc : -sc) * wx * wy;
            j0[0] = fmaf(dx, v[c].x, j0[0]); j0[1] = fmaf(dy, v[c].x, j0[1]); j0[2] = fmaf(dz, v[c].x, j0[2]);
            j1[0] = fmaf(dx, v[c].y, j1[0]); j1[1] = fmaf(dy, v[c].y, j1[1]); j1[2] = fmaf(dz, v[c].y, j1[2]);
        }
    }
    *reinterpret_cast<float2*>(out + i * out_stride + l * 2) = make_float2(a0, a1);
    if (WITH_JAC) {
        float* J = dy_dx + (i * L * 2 + l * 2) * 3;
        J[0] = j0[0]; J[1] = j0[1]; J[2] = j0[2];
        J[3] = j1[0]; J[4] = j1[1]; J[5] = j1[2];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// forward, XCD-PARTITIONED (large batches; since round 2 the default schedule is ONE table per launch with the eight XCD slots
// sharing the points -- see ia_hashgrid_fwd_xcd -- and the per-XCD level plans described here are IA_HASH_XCD_PLAN=passes).
// The level-major kernel above pulls ~4 KB per point through the fabric
// (every 8-byte gather of a fine level drags a 64-byte sector out of the Infinity Cache: measured 18 GB per 4.4 M
// points at 6.2 TB/s, L2 hit rate 0.6), because each XCD's 4 MiB L2 sees all sixteen 4 MB level tables.  Here a
// workgroup's levels are chosen by the XCD it runs on (block b -> XCD b % 8, observed dispatch order; used for speed
// only -- any placement computes the same result): pass A gives each XCD ONE hashed level for all points, pass B
// splits the remaining levels over XCD pairs (half the points each), so the table an XCD gathers from fits its L2.
// Lanes are consecutive points (coalesced 12-byte reads, 8-byte level-major writes); a transpose kernel then builds
// the [n, 2L] rows (and the Jacobian rows) the consumers expect.
struct XcdPlan {
    int first_level[8], n_level[8];     // levels handled by the workgroups of this XCD slot
    int part[8], nparts[8];             // this slot covers points [part, part+1) / nparts of the batch
    int straight;                       // 1 (always): straight-line gather for dense levels and power-of-two hashed tables
};

// General rule of the XCD-partitioned gather (a hashed table that is not a power of two: never the case for tiny-cuda-nn's
// 2^log2_hashmap_size tables): indices through grid_index(), one of three load shapes per x-pair.  This was rounds 1-2's gather.
// It stays compiled into the kernel for a second, unglamorous reason: with this path present hipcc's code for the straight-line
// path below runs 7 % faster (9.6 against 10.3 ms per 50 M points, same-box A/B of the two builds, tools/sdf_head_probe.py with
// IA_AMD_LIB) -- same load bursts in the ISA, another register allocation / block layout; not understood further.
typedef float f4_a8b __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ void load_x_pair(const float2* __restrict__ tab, uint32_t i0, uint32_t i1, float2& v0, float2& v1)
{
    if (i1 == i0 + 1u) {
        const f4_a8b q = *reinterpret_cast<const f4_a8b*>(tab + i0);
        v0 = make_float2(q.x, q.y); v1 = make_float2(q.z, q.w);
    } else if (i0 == i1 + 1u) {
        const f4_a8b q = *reinterpret_cast<const f4_a8b*>(tab + i1);
        v1 = make_float2(q.x, q.y); v0 = make_float2(q.z, q.w);
    } else {
        v0 = tab[i0];
        v1 = tab[i1];
    }
}
template <bool WITH_JAC>
__device__ __forceinline__ void xcd_gather(const float2* __restrict__ tab, uint32_t hsize, uint32_t res, float sc,
                                           const float xs[3], float2 v[8], float pos[3])
{
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float p = fmaf(sc, xs[d], 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)(int)fl;
        pos[d] = p - fl;
    }
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        const uint32_t i0 = grid_index(hsize, res, pg[0], pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1));
        const uint32_t i1 = grid_index(hsize, res, pg[0] + 1u, pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1));
        load_x_pair(tab, i0, i1, v[c], v[c + 1]);
    }
}

// Gather of the XCD-partitioned forward.  The two x-neighbours of a cell edge are adjacent table entries far more often than not:
// always on the dense levels (index = x + y res + z res^2) and, on the hashed levels, whenever x is even (x + 1 = x ^ 1 only flips
// bit 0 of x ^ y P1 ^ z P2), so an x-pair is one ALIGNED 16-byte load of the entry pair that holds i0 -- it also holds i1 whenever
// i1 == i0 ^ 1 -- plus one 8-byte load of i1 (redundant in that case: same line), all UNCONDITIONAL.  (Rounds 1-2 picked one of
// three load shapes per lane, one 16-byte load where the pair is adjacent and two 8-byte loads otherwise: a third fewer requests, but
// hipcc compiles that into divergent blocks whose results meet in the same registers and ends every block with s_waitcnt vmcnt(0)
// -- the ISA read "load load wait, load wait, load wait": one or two gathers in flight per lane on a kernel that is bound by gather
// latency x requests in flight.  Straight-line, the 16 loads of a lane's two points are issued back to back and waited for once:
// 10.7 -> 9.6 ms per 50 M points, 83.5 -> 74.7 ms per headline step.)
// Two points of one level, straight-line: the level's index rule is chosen ONCE (wave-uniform template argument) instead of
// inside grid_index(), so nothing splits the basic block -- 16 index computations, then 16 loads back to back, then the selects.
//   HASHED: tiny-cuda-nn's coherent prime hash masked to the power-of-two table.
//   dense : x + y res + z res^2, reduced modulo the table size by one conditional subtraction (the index is below twice the
//           table size: res^3 <= hsize and every coordinate is at most res).
template <bool HASHED>
__device__ __forceinline__ uint32_t level_index(uint32_t hsize, uint32_t res, uint32_t px, uint32_t py, uint32_t pz)
{
    if (HASHED) return ((px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u)) & (hsize - 1u);
    const uint32_t index = px + res * (py + res * pz);
    return min(index >= hsize ? index - hsize : index, hsize - 1u);      // the clamp only bites for points outside the unit cube (redone below)
}

template <bool HASHED>
__device__ __forceinline__ void xcd_gather2(const float2* __restrict__ tab, uint32_t hsize, uint32_t res, float sc, const float xa[3],
                                            const float xb[3], float2 va[8], float2 vb[8], float pa[3], float pb[3])
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float f4_a16 __attribute__((ext_vector_type(4), aligned(16)));
    uint32_t ga[3], gb[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float p = fmaf(sc, xa[d], 0.5f), fl = floorf(p);
        ga[d] = (uint32_t)(int)fl; pa[d] = p - fl;
        const float p2 = fmaf(sc, xb[d], 0.5f), fl2 = floorf(p2);
        gb[d] = (uint32_t)(int)fl2; pb[d] = p2 - fl2;
    }
    uint32_t i0[8], i1[8];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        i0[c] = level_index<HASHED>(hsize, res, ga[0], ga[1] + (c & 1), ga[2] + ((c >> 1) & 1));
        i1[c] = level_index<HASHED>(hsize, res, ga[0] + 1u, ga[1] + (c & 1), ga[2] + ((c >> 1) & 1));
        i0[4 + c] = level_index<HASHED>(hsize, res, gb[0], gb[1] + (c & 1), gb[2] + ((c >> 1) & 1));
        i1[4 + c] = level_index<HASHED>(hsize, res, gb[0] + 1u, gb[1] + (c & 1), gb[2] + ((c >> 1) & 1));
    }
    f4_a16 q[8];
    v2f t[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        q[c] = *reinterpret_cast<const f4_a16*>(tab + (i0[c] & ~1u));
        t[c] = *reinterpret_cast<const v2f*>(tab + i1[c]);
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const bool odd = (i0[c] & 1u) != 0u;
        const float2 lo = make_float2(q[c].x, q[c].y), hi = make_float2(q[c].z, q[c].w);
        const float2 e0 = odd ? hi : lo;
        const float2 e1 = (i1[c] == (i0[c] ^ 1u)) ? (odd ? lo : hi) : make_float2(t[c].x, t[c].y);
        if (c < 4) { va[2 * c] = e0; va[2 * c + 1] = e1; } else { vb[2 * (c - 4)] = e0; vb[2 * (c - 4) + 1] = e1; }
    }
    if (!HASHED) {
        // a point outside the unit cube (a candidate of the search outside the field's box) has wrapped cell coordinates: its
        // index needs the full modulo of grid_index(), as tiny-cuda-nn computes it.  Rare and divergent: redone here, after the
        // loads of the common case.
        if (ga[0] > res || ga[1] > res || ga[2] > res)
#pragma unroll
            for (int c = 0; c < 8; c++) va[c] = tab[grid_index(hsize, res, ga[0] + (c & 1), ga[1] + ((c >> 1) & 1), ga[2] + ((c >> 2) & 1))];
        if (gb[0] > res || gb[1] > res || gb[2] > res)
#pragma unroll
            for (int c = 0; c < 8; c++) vb[c] = tab[grid_index(hsize, res, gb[0] + (c & 1), gb[1] + ((c >> 1) & 1), gb[2] + ((c >> 2) & 1))];
    }
}

template <bool WITH_JAC>
__device__ __forceinline__ void xcd_blend_store(const float2 v[8], const float pos[3], float sc, int64_t slot,
                                                float2* __restrict__ tmp, float* __restrict__ tmp_jac)
{
    float a0 = 0.f, a1 = 0.f;
    float j0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float wx = (c & 1) ? pos[0] : 1.0f - pos[0];
        const float wy = (c & 2) ? pos[1] : 1.0f - pos[1];
        const float wz = (c & 4) ? pos[2] : 1.0f - pos[2];
        const float w = wx * wy * wz;
        a0 = fmaf(w, v[c].x, a0);       // explicit: the flat and the level-major kernels must round alike whatever hipcc contracts
        a1 = fmaf(w, v[c].y, a1);
        if (WITH_JAC) {
            const float dx = ((c & 1) ? sc : -sc) * wy * wz;
            const float dy = ((c & 2) ? sc : -sc) * wx * wz;
            const float dz = ((c & 4) ? sc : -sc) * wx * wy;
            j0[0] = fmaf(dx, v[c].x, j0[0]); j0[1] = fmaf(dy, v[c].x, j0[1]); j0[2] = fmaf(dz, v[c].x, j0[2]);
            j1[0] = fmaf(dx, v[c].y, j1[0]); j1[1] = fmaf(dy, v[c].y, j1[1]); j1[2] = fmaf(dz, v[c].y, j1[2]);
        }
    }
    // streaming outputs bypass the caches' retention (non-temporal): the L2 should hold the level's 4 MB table, nothing else
    // (non-temporal GATHERS of the table itself were measured too, for the levels from 13 / 9 / 5 / 0 up: 18 / 31 / 33 / 37 % slower)
    typedef float v2f __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store((v2f){a0, a1}, reinterpret_cast<v2f*>(tmp + slot));
    if (WITH_JAC) {
        v2f* J = reinterpret_cast<v2f*>(tmp_jac + slot * 6);
        __builtin_nontemporal_store((v2f){j0[0], j0[1]}, J);
        __builtin_nontemporal_store((v2f){j0[2], j1[0]}, J + 1);
        __builtin_nontemporal_store((v2f){j1[1], j1[2]}, J + 2);
    }
}

template <bool WITH_JAC>
__global__ __launch_bounds__(THREADS) void hash_fwd_xcd_kernel(int64_t n, const float* __restrict__ x,
                                                                const float2* __restrict__ params, HashCfg cfg, XcdPlan plan,
                                                                float2* __restrict__ tmp /*[L][n]*/,
                                                                float* __restrict__ tmp_jac /*[L][n][6]*/)
{
    const int slot = blockIdx.x & 7;
    const int64_t chunk = blockIdx.x >> 3, nchunks = gridDim.x >> 3;
    const int np = plan.nparts[slot];
    const int64_t per = (n + np - 1) / np;
    const int64_t lo = per * plan.part[slot], hi = (lo + per < n) ? lo + per : n;
    const int64_t stride = nchunks * THREADS;
    // two points per lane and iteration: the gathers of both (16 independent 8-byte loads) are issued before either
    // blend -- the kernel is bound by L2 gather latency x requests in flight, not by bandwidth (L2 hit 0.91)
    for (int64_t i = lo + chunk * THREADS + threadIdx.x; i < hi; i += 2 * stride) {
        const int64_t i2 = i + stride;
        const bool two = i2 < hi;
        const int64_t ib = two ? i2 : i;
        const float xa[3] = {__builtin_nontemporal_load(x + i * 3 + 0), __builtin_nontemporal_load(x + i * 3 + 1),
                             __builtin_nontemporal_load(x + i * 3 + 2)};
        const float xb[3] = {__builtin_nontemporal_load(x + ib * 3 + 0), __builtin_nontemporal_load(x + ib * 3 + 1),
                             __builtin_nontemporal_load(x + ib * 3 + 2)};
        for (int l = plan.first_level[slot]; l < plan.first_level[slot] + plan.n_level[slot]; l++) {
            const float sc = cfg.scale[l];
            const uint32_t res = cfg.res[l], hsize = cfg.offsets[l + 1] - cfg.offsets[l];
            const float2* tab = params + cfg.offsets[l];
            float2 va[8], vb[8];
            float pa[3], pb[3];
            // grid_index()'s rule for this level (wave-uniform): dense while res^3 fits the table, else hashed (power-of-two table)
            const uint64_t r64 = res;
            const bool hashed = r64 * r64 * r64 > (uint64_t)hsize;
            if (plan.straight && hashed && (hsize & (hsize - 1u)) == 0u) xcd_gather2<true>(tab, hsize, res, sc, xa, xb, va, vb, pa, pb);
            else if (plan.straight && !hashed) xcd_gather2<false>(tab, hsize, res, sc, xa, xb, va, vb, pa, pb);
            else {
                xcd_gather<WITH_JAC>(tab, hsize, res, sc, xa, va, pa);
                xcd_gather<WITH_JAC>(tab, hsize, res, sc, xb, vb, pb);
            }
            xcd_blend_store<WITH_JAC>(va, pa, sc, (int64_t)l * n + i, tmp, tmp_jac);
            if (two) xcd_blend_store<WITH_JAC>(vb, pb, sc, (int64_t)l * n + i2, tmp, tmp_jac);
        }
    }
}

// level-major [L][n] -> rows [n, 2L] (+ Jacobian [L][n][6] -> [n, 2L, 3]); one workgroup = 256 points, through LDS
template <bool WITH_JAC>
__global__ __launch_bounds__(THREADS) void hash_transpose_kernel(int64_t n, int L, const float2* __restrict__ tmp,
                                                                  const float* __restrict__ tmp_jac, float* __restrict__ out,
                                                                  int out_stride, float* __restrict__ dy_dx)
{
    extern __shared__ __attribute__((aligned(16))) float s_t[];      // [256][2L+1]  (and [256][6L+1] for the Jacobian)
    const int64_t p0 = (int64_t)blockIdx.x * THREADS;
    const int rows = (int)((n - p0) < THREADS ? (n - p0) : THREADS);
    const int ld = 2 * L + 1;
    for (int l = 0; l < L; l++)
        if ((int)threadIdx.x < rows) {
            const float2 v = tmp[(int64_t)l * n + p0 + threadIdx.x];
            s_t[threadIdx.x * ld + 2 * l] = v.x; s_t[threadIdx.x * ld + 2 * l + 1] = v.y;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * 2 * L; i += THREADS) {
        const int r = i / (2 * L), c = i % (2 * L);
        out[(p0 + r) * out_stride + c] = s_t[r * ld + c];
    }
    if (WITH_JAC) {
        __syncthreads();
        const int ldj = 6 * L + 1;
        for (int l = 0; l < L; l++)
            if ((int)threadIdx.x < rows) {
                const float* J = tmp_jac + ((int64_t)l * n + p0 + threadIdx.x) * 6;
#pragma unroll
                for (int k = 0; k < 6; k++) s_t[threadIdx.x * ldj + 6 * l + k] = J[k];
            }
        __syncthreads();
        for (int i = threadIdx.x; i < rows * 6 * L; i += THREADS) {
            const int r = i / (6 * L), c = i % (6 * L);
            dy_dx[(p0 + r) * 6 * L + c] = s_t[r * ldj + c];
        }
    }
}

// backward w.r.t. the table:
//   grad[c] += gE[l,:] * w_c  +  gG[l,:] * sum_a q[a] * d w_c / d x_a      (second term optional)
//
// backward w.r.t. the table, run-merged: one lane per POINT, loop over levels and corners.  Samples arrive
// in marching order, so consecutive lanes very often fall into the same cell (always at the coarse
// levels, where a plain atomic scatter serialises on a few thousand addresses).  Equal consecutive
// addresses are merged with a wave-level segmented scan and only the last lane of each run issues
// the atomics: 64x fewer atomics on the coarse levels, unchanged on the finest, identical result up to
// summation order.
template <bool SECOND>
__global__ __launch_bounds__(THREADS) void hash_bwd_runs_kernel(int64_t n, const float* __restrict__ x, HashCfg cfg,
                                                                 const float* __restrict__ gE, int gE_stride,
                                                                 const float* __restrict__ gG, int gG_stride,
                                                                 const float* __restrict__ q, float* __restrict__ grad)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = i < n;
    const int64_t ii = active ? i : n - 1;
    const float x0 = x[ii * 3 + 0], x1 = x[ii * 3 + 1], x2 = x[ii * 3 + 2];
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (SECOND) { qx = q[ii * 3 + 0]; qy = q[ii * 3 + 1]; qz = q[ii * 3 + 2]; }
    for (int l = 0; l < cfg.n_levels; l++) {
        const float sc = cfg.scale[l];
        const uint32_t res = cfg.res[l], hsize = cfg.offsets[l + 1] - cfg.offsets[l];
        float* tab = grad + (int64_t)cfg.offsets[l] * 2;
        float2 e = make_float2(0.f, 0.f), g = make_float2(0.f, 0.f);
        if (active) {
            if (gE) e = *reinterpret_cast<const float2*>(gE + i * gE_stride + l * 2);
            if (SECOND) g = *reinterpret_cast<const float2*>(gG + i * gG_stride + l * 2);
        }
        const bool any_here = (e.x != 0.f) || (e.y != 0.f) || (g.x != 0.f) || (g.y != 0.f);
        if (!__any(any_here)) continue;                       // masked-out level for the whole wave
        float pos[3];
        uint32_t pg[3];
        {
            const float xs[3] = {x0, x1, x2};
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float p = fmaf(sc, xs[d], 0.5f);
                const float fl = floorf(p);
                pg[d] = (uint32_t)(int)fl;
                pos[d] = p - fl;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float wx = (c & 1) ? pos[0] : 1.0f - pos[0];
            const float wy = (c & 2) ? pos[1] : 1.0f - pos[1];
            const float wz = (c & 4) ? pos[2] : 1.0f - pos[2];
            const float w0 = wx * wy * wz;
            float vx = e.x * w0, vy = e.y * w0;
            if (SECOND) {
                const float dw = ((c & 1) ? sc : -sc) * wy * wz * qx + ((c & 2) ? sc : -sc) * wx * wz * qy +
                                 ((c & 4) ? sc : -sc) * wx * wy * qz;
                vx += g.x * dw;
                vy += g.y * dw;
            }
            uint32_t idx = grid_index(hsize, res, pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1));
            if (!active) { idx = 0xFFFFFFFFu; vx = 0.f; vy = 0.f; }
            // run heads: lane 0 or address differs from the previous lane
            const uint32_t prev = __shfl_up(idx, 1, 64);
            const bool head = (lane == 0) || (prev != idx);
            const unsigned long long heads = __ballot(head);
            // position of this lane's run head = highest set bit of heads at or below `lane`
            const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
            const int hpos = 63 - __clzll(below);
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float ax = __shfl_up(vx, off, 64), ay = __shfl_up(vy, off, 64);
                if (lane - off >= hpos) { vx += ax; vy += ay; }
            }
            const bool tail = (lane == 63) || ((heads >> (lane + 1)) & 1ull);
            if (tail && active && (vx != 0.f || vy != 0.f)) {
                unsafeAtomicAdd(tab + (int64_t)idx * 2 + 0, vx);
                unsafeAtomicAdd(tab + (int64_t)idx * 2 + 1, vy);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// backward w.r.t. the table, BINNED (multisplit + LDS reduction) -- the default for large batches.
//
// A device-scope fp32 atomic costs the same wherever it lands (they execute at the memory side of the fabric:
// ~21 G atomics/s measured on MI355X), so the scatter above is pinned at n*L*8*2 atomics no matter how it is
// scheduled.  This path issues (almost) none.  The table of every level is cut into slices of 8192 entries (64 KB of
// float2 -- one LDS-resident accumulator per workgroup); a "bucket" is one (level, slice):
//   1. count : per workgroup (1024 points) an LDS histogram of its records per bucket -> counts[bucket][wg]
//   2. scan  : exclusive scan of that matrix in bucket-major order = the base of every (bucket, wg) run
//   3. fill  : records (13-bit local index as u16, float2 value) are written to their run (wave-level run merging
//              first: consecutive samples of a ray falling into the same cell become one record)
//   4. reduce: one workgroup per (bucket, part) streams its records (coalesced 10 B / record), accumulates them
//              with LDS atomics (ds_add_f32), and adds the slice to the gradient table -- plain read-modify-write
//              when it owns the whole bucket, atomics only for the few buckets split in parts (coarse dense levels).
// HBM traffic ~ 2 x 10 B per record instead of 2 fabric atomics per record.
constexpr int SLICE_LOG2 = 13;
constexpr int SLICE = 1 << SLICE_LOG2;
constexpr int BIN_ROUNDS = 4;
constexpr int BIN_TILE = THREADS * BIN_ROUNDS;      // points per workgroup
constexpr int MAX_BUCKETS = 1024;
constexpr int MAX_PARTS = 64;
constexpr int PART_RECORDS = 1 << 19;

struct BinCfg {
    int n_buckets;
    int bstart[MAX_LEVELS + 1];
};

__host__ void make_bins(BinCfg& b, const HashCfg& c)
{
    int k = 0;
    for (int l = 0; l < c.n_levels; l++) {
        b.bstart[l] = k;
        k += (int)((c.offsets[l + 1] - c.offsets[l] + SLICE - 1) >> SLICE_LOG2);
    }
    b.bstart[c.n_levels] = k;
    b.n_buckets = k;
}

// running maximum of a level's |record value| (scale of the fixed-point reduction).  Every wave used to fire one
// device-scope atomicMax per level at the SAME address: ~400 k same-address atomics serialise at the memory side and
// cost 1.7 - 3.5 ms per launch (measured by ablation).  The maximum converges after a few waves, so look first (a stale
// value only means one redundant atomic) and touch the address only when this wave would raise it.
__device__ __forceinline__ void level_max_update(unsigned* p, float lmax)
{
    const unsigned v = __float_as_uint(lmax);                 // non-negative floats order like their bit patterns
    if (v > __atomic_load_n(p, __ATOMIC_RELAXED)) atomicMax(p, v);
}

template <bool SECOND, bool FILL>
__global__ __launch_bounds__(THREADS) void hash_bin_kernel(int64_t n, const float* __restrict__ x, HashCfg cfg, BinCfg bins,
                                                            const float* __restrict__ gE, int gE_stride,
                                                            const float* __restrict__ gG, int gG_stride,
                                                            const float* __restrict__ q, int nwg,
                                                            int32_t* __restrict__ counts, uint16_t* __restrict__ rec_idx,
                                                            float2* __restrict__ rec_val, uint32_t level_mask,
                                                            unsigned* __restrict__ level_max)
{
    // Loop order: level-major over the workgroup's 1024 points (4 per lane), so that at any time a workgroup appends to
    // the <= 64 runs of ONE level only (64 x 2 open cache lines instead of 771 x 2): the scattered 2- and 8-byte record
    // stores then merge in L2 into full lines before they are evicted to HBM.  Gradient rows are fetched 4 levels at a
    // time (one 32-byte piece per point and array), positions once.
    __shared__ int hist[MAX_BUCKETS];
    __shared__ int base[FILL ? MAX_BUCKETS : 1];
    const int lane = threadIdx.x & 63;
    for (int b = threadIdx.x; b < bins.n_buckets; b += THREADS) {
        hist[b] = 0;
        if (FILL) base[b] = counts[(int64_t)b * nwg + blockIdx.x];
    }
    __syncthreads();
    float px[BIN_ROUNDS][3], pq[BIN_ROUNDS][3];
    bool act[BIN_ROUNDS];
    int64_t pi[BIN_ROUNDS];
#pragma unroll
    for (int r = 0; r < BIN_ROUNDS; r++) {
        const int64_t i = (int64_t)blockIdx.x * BIN_TILE + r * THREADS + threadIdx.x;
        act[r] = i < n;
        pi[r] = act[r] ? i : n - 1;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            px[r][d] = x[pi[r] * 3 + d];
            pq[r][d] = (SECOND && FILL) ? q[pi[r] * 3 + d] : 0.0f;
        }
    }
    constexpr int LG = 4;                                     // levels per gradient fetch
    for (int l0 = 0; l0 < cfg.n_levels; l0 += LG) {
        float2 ev[BIN_ROUNDS][LG], gv[BIN_ROUNDS][LG];
#pragma unroll
        for (int r = 0; r < BIN_ROUNDS; r++)
#pragma unroll
            for (int j = 0; j < LG; j++) {
                ev[r][j] = make_float2(0.f, 0.f);
                gv[r][j] = make_float2(0.f, 0.f);
                if (FILL && act[r] && l0 + j < cfg.n_levels && ((level_mask >> (l0 + j)) & 1u)) {
                    if (gE) ev[r][j] = *reinterpret_cast<const float2*>(gE + pi[r] * gE_stride + (l0 + j) * 2);
                    if (SECOND) gv[r][j] = *reinterpret_cast<const float2*>(gG + pi[r] * gG_stride + (l0 + j) * 2);
                }
            }
#pragma unroll
        for (int j = 0; j < LG; j++) {
            const int l = l0 + j;
            if (l >= cfg.n_levels) break;
            if (!((level_mask >> l) & 1u)) continue;              // level switched off by the caller (progressive bands)
            float lmax = 0.0f;
            const float sc = cfg.scale[l];
            const uint32_t res = cfg.res[l], hsize = cfg.offsets[l + 1] - cfg.offsets[l];
            const int b0 = bins.bstart[l];
#pragma unroll
            for (int r = 0; r < BIN_ROUNDS; r++) {
                const float2 e = ev[r][j], g = gv[r][j];
                const bool active = act[r];
                float pos[3];
                uint32_t pg[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const float p = fmaf(sc, px[r][d], 0.5f);
                    const float fl = floorf(p);
                    pg[d] = (uint32_t)(int)fl;
                    pos[d] = p - fl;
                }
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    uint32_t idx = grid_index(hsize, res, pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1));
                    if (!active) idx = 0xFFFFFFFFu;
                    const uint32_t prev = __shfl_up(idx, 1, 64);
                    const bool head = (lane == 0) || (prev != idx);
                    const unsigned long long heads = __ballot(head);
                    const bool tail = (lane == 63) || ((heads >> (lane + 1)) & 1ull);
                    float vx = 0.f, vy = 0.f;
                    if (FILL) {
                        const float wx = (c & 1) ? pos[0] : 1.0f - pos[0];
                        const float wy = (c & 2) ? pos[1] : 1.0f - pos[1];
                        const float wz = (c & 4) ? pos[2] : 1.0f - pos[2];
                        const float w0 = wx * wy * wz;
                        vx = e.x * w0; vy = e.y * w0;
                        if (SECOND) {
                            const float dw = ((c & 1) ? sc : -sc) * wy * wz * pq[r][0] + ((c & 2) ? sc : -sc) * wx * wz * pq[r][1] +
                                             ((c & 4) ? sc : -sc) * wx * wy * pq[r][2];
                            vx += g.x * dw;
                            vy += g.y * dw;
                        }
                        if (!active) { vx = 0.f; vy = 0.f; }
                        const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
                        const int hpos = 63 - __clzll(below);
                        if (heads != ~0ull) {                           // wave-uniform: some run is longer than one lane
#pragma unroll
                            for (int off = 1; off < 64; off <<= 1) {
                                const float ax = __shfl_up(vx, off, 64), ay = __shfl_up(vy, off, 64);
                                if (lane - off >= hpos) { vx += ax; vy += ay; }
                            }
                        }
                    }
                    if (tail && active) {
                        const int b = b0 + (int)(idx >> SLICE_LOG2);
                        const int rank = atomicAdd(&hist[b], 1);
                        if (FILL) {
                            lmax = fmaxf(lmax, fmaxf(fabsf(vx), fabsf(vy)));
                            const int64_t p = (int64_t)base[b] + rank;
                            rec_idx[p] = (uint16_t)(idx & (SLICE - 1));
                            rec_val[p] = make_float2(vx, vy);
                        }
                    }
                }
            }
            if (FILL) {                                           // per-level max |value| -> fixed-point scale of the reduction
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
                if (lane == 0 && lmax > 0.0f) level_max_update(level_max + l, lmax);
            }
        }
    }
    if (!FILL) {
        __syncthreads();
        for (int b = threadIdx.x; b < bins.n_buckets; b += THREADS) counts[(int64_t)b * nwg + blockIdx.x] = hist[b];
    }
}


// ---- binned backward, STAGED variant: wave-private units and full-line record stores ---------------------------------
// The fill above appends every record straight to its (bucket, workgroup) run: a wave's store instruction scatters 64
// lanes over up to 64 runs (PMC: 6.7 GB written + 3-6 GB write-allocated for 4.2 GB of records).  Here the unit of the
// count -> scan -> fill protocol is ONE WAVE (128 points): per level the wave sorts its <= 1024 records by slice in a
// 10 KB LDS staging area (offsets come from the scanned counts, positions from integer LDS atomics), then copies the
// sorted block out so that consecutive lanes write consecutive records of a run.  No workgroup barrier anywhere: waves
// never share state (LDS operations of one wave execute in order).
constexpr int U_ROUNDS = 2;
constexpr int U_PTS = 64 * U_ROUNDS;                  // points per unit (wave)
constexpr int U_REC = U_PTS * 8;                      // records per unit and level
constexpr int U_WAVES = THREADS / 64;

struct UnitLds {
    unsigned long long cur[64];      // per slice: (global run start - staging offset) << 32 | next staging slot
    float4 rec[U_REC];               // staged records, sorted by slice: (value.x, value.y, bits(destination), bits(local index))
};

template <bool SECOND, bool FILL>
__global__ __launch_bounds__(THREADS) void hash_bin_unit_kernel(int64_t n, const float* __restrict__ x, HashCfg cfg, BinCfg bins,
                                                                 const float* __restrict__ gE, int gE_stride,
                                                                 const float* __restrict__ gG, int gG_stride,
                                                                 const float* __restrict__ q, int nunits, int64_t m,
                                                                 int32_t* __restrict__ counts, const int32_t* __restrict__ total,
                                                                 uint16_t* __restrict__ rec_idx, float2* __restrict__ rec_val,
                                                                 uint32_t level_mask, unsigned* __restrict__ level_max)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.x * U_WAVES + wave;
    int* hist = reinterpret_cast<int*>(smem_raw) + wave * MAX_BUCKETS;                 // COUNT: per-wave histogram
    UnitLds& U = reinterpret_cast<UnitLds*>(smem_raw)[wave];                           // FILL : per-wave staging
    if (!FILL)
        for (int b = lane; b < bins.n_buckets; b += 64) hist[b] = 0;
    float px[U_ROUNDS][3], pq[U_ROUNDS][3];
    bool act[U_ROUNDS];
    int64_t pi[U_ROUNDS];
#pragma unroll
    for (int r = 0; r < U_ROUNDS; r++) {
        const int64_t i = (int64_t)unit * U_PTS + r * 64 + lane;
        act[r] = i < n;
        pi[r] = act[r] ? i : n - 1;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            px[r][d] = x[pi[r] * 3 + d];
            pq[r][d] = (SECOND && FILL) ? q[pi[r] * 3 + d] : 0.0f;
        }
    }
    constexpr int LG = 4;
    for (int l0 = 0; l0 < cfg.n_levels; l0 += LG) {
        // gradient rows: two levels (16 bytes) per load -- every lane reads its own row, so a load instruction costs 64
        // line accesses however wide it is (tools/probes/tcp_mask_probe.hip); half as many of them
        float2 ev[U_ROUNDS][LG], gv[U_ROUNDS][LG];
#pragma unroll
        for (int r = 0; r < U_ROUNDS; r++)
#pragma unroll
            for (int j = 0; j < LG; j += 2) {
                ev[r][j] = ev[r][j + 1] = make_float2(0.f, 0.f);
                gv[r][j] = gv[r][j + 1] = make_float2(0.f, 0.f);
                const int l = l0 + j;
                if (FILL && act[r] && l + 1 < cfg.n_levels && ((level_mask >> l) & 3u)) {
                    if (gE) {
                        const f4_a8 t = *reinterpret_cast<const f4_a8*>(gE + pi[r] * gE_stride + l * 2);
                        ev[r][j] = make_float2(t.x, t.y); ev[r][j + 1] = make_float2(t.z, t.w);
                    }
                    if (SECOND) {
                        const f4_a8 t = *reinterpret_cast<const f4_a8*>(gG + pi[r] * gG_stride + l * 2);
                        gv[r][j] = make_float2(t.x, t.y); gv[r][j + 1] = make_float2(t.z, t.w);
                    }
                } else if (FILL && act[r] && l < cfg.n_levels && ((level_mask >> l) & 1u)) {      // odd level count: last level
                    if (gE) ev[r][j] = *reinterpret_cast<const float2*>(gE + pi[r] * gE_stride + l * 2);
                    if (SECOND) gv[r][j] = *reinterpret_cast<const float2*>(gG + pi[r] * gG_stride + l * 2);
                }
            }
#pragma unroll
        for (int j = 0; j < LG; j++) {
            const int l = l0 + j;
            if (l >= cfg.n_levels) break;
            if (!((level_mask >> l) & 1u)) continue;
            float lmax = 0.0f;
            const float sc = cfg.scale[l];
            const uint32_t res = cfg.res[l], hsize = cfg.offsets[l + 1] - cfg.offsets[l];
            const int b0 = bins.bstart[l], nb = bins.bstart[l + 1] - b0;
            int tot = 0;
            if (FILL) {
                // this unit's run of every slice of the level: start and length from the scanned count matrix
                int cnt = 0, start = 0;
                if (lane < nb && unit < nunits) {
                    const int64_t flat = (int64_t)(b0 + lane) * nunits + unit;
                    start = counts[flat];
                    cnt = ((flat + 1 < m) ? counts[flat + 1] : total[0]) - start;
                }
                int incl = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                tot = __shfl(incl, 63, 64);
                const int off = incl - cnt;
                U.cur[lane] = ((unsigned long long)(unsigned)(start - off) << 32) | (unsigned)off;
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int r = 0; r < U_ROUNDS; r++) {
                const float2 e = ev[r][j], g = gv[r][j];
                const bool active = act[r];
                float pos[3];
                uint32_t pg[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const float p = fmaf(sc, px[r][d], 0.5f);
                    const float fl = floorf(p);
                    pg[d] = (uint32_t)(int)fl;
                    pos[d] = p - fl;
                }
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    uint32_t idx = grid_index(hsize, res, pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1));
                    if (!active) idx = 0xFFFFFFFFu;
                    const uint32_t prev = __shfl_up(idx, 1, 64);
                    const bool head = (lane == 0) || (prev != idx);
                    const unsigned long long heads = __ballot(head);
                    const bool tail = (lane == 63) || ((heads >> (lane + 1)) & 1ull);
                    float vx = 0.f, vy = 0.f;
                    if (FILL) {
                        const float wx = (c & 1) ? pos[0] : 1.0f - pos[0];
                        const float wy = (c & 2) ? pos[1] : 1.0f - pos[1];
                        const float wz = (c & 4) ? pos[2] : 1.0f - pos[2];
                        const float w0 = wx * wy * wz;
                        vx = e.x * w0; vy = e.y * w0;
                        if (SECOND) {
                            const float dw = ((c & 1) ? sc : -sc) * wy * wz * pq[r][0] + ((c & 2) ? sc : -sc) * wx * wz * pq[r][1] +
                                             ((c & 4) ? sc : -sc) * wx * wy * pq[r][2];
                            vx += g.x * dw;
                            vy += g.y * dw;
                        }
                        if (!active) { vx = 0.f; vy = 0.f; }
                        const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
                        const int hpos = 63 - __clzll(below);
                        if (heads != ~0ull) {
#pragma unroll
                            for (int off = 1; off < 64; off <<= 1) {
                                const float ax = __shfl_up(vx, off, 64), ay = __shfl_up(vy, off, 64);
                                if (lane - off >= hpos) { vx += ax; vy += ay; }
                            }
                        }
                    }
                    if (tail && active) {
                        const int bin = (int)(idx >> SLICE_LOG2);
                        if (FILL) {
                            lmax = fmaxf(lmax, fmaxf(fabsf(vx), fabsf(vy)));
                            // one 64-bit LDS atomic hands out the staging slot AND the slice's destination delta
                            const unsigned long long t = atomicAdd(&U.cur[bin], 1ull);
                            const unsigned slot = (unsigned)t & (U_REC - 1);
                            const unsigned d = (unsigned)(t >> 32) + (unsigned)t;
                            U.rec[slot] = make_float4(vx, vy, __uint_as_float(d), __uint_as_float(idx & (SLICE - 1)));
                        } else {
                            atomicAdd(&hist[b0 + bin], 1);
                        }
                    }
                }
            }
            if (FILL) {
                __builtin_amdgcn_wave_barrier();
                // copy-out: the staging block is sorted by slice, so consecutive lanes write consecutive records of a run
                for (int k = lane; k < tot; k += 64) {
                    const float4 rcd = U.rec[k];
                    const unsigned d = __float_as_uint(rcd.z);
                    rec_idx[d] = (uint16_t)__float_as_uint(rcd.w);
                    rec_val[d] = make_float2(rcd.x, rcd.y);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
                if (lane == 0 && lmax > 0.0f) level_max_update(level_max + l, lmax);
            }
        }
    }
    if (!FILL && unit < nunits)
        for (int b = lane; b < bins.n_buckets; b += 64) counts[(int64_t)b * nunits + unit] = hist[b];
}

// LDS accumulation in 64-bit FIXED POINT with integer atomics.  Measured on MI355X (tools/probes/lds_atomic_probe.hip):
// ds_add_f32 retires 0.17 updates / clk / CU, ds_add_u64 5.6 -- the float LDS atomic is ~30x slower than the integer
// one, and it alone made this kernel 6x slower than its HBM-read time.  Each level gets a power-of-two scale from the
// largest |value| of its records (2^40 / 2^ceil(log2 max)): every record is converted with at most 2^-40 max relative
// quantisation (exact for |v| >= 2^-16 max), the sums themselves are exact integer sums -- order independent: a bucket
// reduced in one part is bit-reproducible run to run (buckets split in parts still meet in float atomics below).
constexpr int RTHREADS = 512;
__device__ __forceinline__ void fx_add(unsigned long long* a, uint32_t i, float vx, float vy, float scale)
{
    atomicAdd(a + 2 * i, (unsigned long long)__float2ll_rn(vx * scale));
    atomicAdd(a + 2 * i + 1, (unsigned long long)__float2ll_rn(vy * scale));
}

// work plan of the reduction: parts per bucket (buckets larger than PART_RECORDS are split), exclusive prefix in
// plan[0 .. n_buckets], work-item counter in plan[n_buckets + 1].  One workgroup.
__global__ __launch_bounds__(1024) void hash_plan_kernel(BinCfg bins, int nwg, const int32_t* __restrict__ scan,
                                                          const int32_t* __restrict__ total, int32_t* __restrict__ plan)
{
    __shared__ int parts[MAX_BUCKETS];
    const int b = threadIdx.x;
    int np = 0;
    if (b < bins.n_buckets) {
        const int start = scan[(int64_t)b * nwg];
        const int end = (b + 1 < bins.n_buckets) ? scan[(int64_t)(b + 1) * nwg] : *total;
        const int cnt = end - start;
        np = cnt <= 0 ? 0 : (cnt + PART_RECORDS - 1) / PART_RECORDS;
        np = np > MAX_PARTS ? MAX_PARTS : np;
    }
    parts[b] = np;
    __syncthreads();
    if (b == 0) {
        int run = 0;
        for (int k = 0; k < bins.n_buckets; k++) { plan[k] = run; run += parts[k]; }
        plan[bins.n_buckets] = run;
        plan[bins.n_buckets + 1] = 0;
    }
}

// persistent: one workgroup per CU pulls (bucket, part) items off the plan
__global__ __launch_bounds__(RTHREADS) void hash_reduce_kernel(HashCfg cfg, BinCfg bins, int nwg,
                                                                const int32_t* __restrict__ scan,
                                                                const int32_t* __restrict__ total,
                                                                const uint16_t* __restrict__ rec_idx,
                                                                const float2* __restrict__ rec_val,
                                                                const unsigned* __restrict__ level_max,
                                                                int32_t* __restrict__ plan, float* __restrict__ grad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];      // [SLICE][2]
    __shared__ int s_item;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_item = atomicAdd(plan + bins.n_buckets + 1, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= plan[bins.n_buckets]) return;
    int lo = 0, hi = bins.n_buckets;                                              // last bucket with plan[b] <= item
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (plan[mid] <= item) lo = mid; else hi = mid; }
    const int b = lo, part = item - plan[lo];
    const int start = scan[(int64_t)b * nwg];
    const int end = (b + 1 < bins.n_buckets) ? scan[(int64_t)(b + 1) * nwg] : *total;
    const int cnt = end - start;
    int nparts = (cnt + PART_RECORDS - 1) / PART_RECORDS;
    nparts = nparts > MAX_PARTS ? MAX_PARTS : nparts;
    const int per = (cnt + nparts - 1) / nparts;
    const int s = start + part * per;
    const int e = (s + per < end) ? s + per : end;
    int l = 0;
    while (b >= bins.bstart[l + 1]) l++;
    const float lmax = __uint_as_float(level_max[l]);
    if (!(lmax > 0.0f) || !(lmax < 3.0e38f)) continue;          // all-zero level (or non-finite gradients: nothing sane to add)
    int ex;
    (void)frexpf(lmax, &ex);                                    // lmax < 2^ex
    const float scale = ldexpf(1.0f, 40 - ex), inv_scale = ldexpf(1.0f, ex - 40);
    const int slice = b - bins.bstart[l];
    const uint32_t hsize = cfg.offsets[l + 1] - cfg.offsets[l];
    const int first = slice << SLICE_LOG2;
    const int entries = ((int)hsize - first < SLICE) ? (int)hsize - first : SLICE;
    for (int k = threadIdx.x; k < 2 * entries; k += RTHREADS) acc[k] = 0ull;
    __syncthreads();
    // head up to the first multiple of 4, then 4 consecutive records per lane (8-byte index load, 2 x 16-byte value
    // loads), two groups in flight per lane; tail one by one.
    const int s4 = (s + 3) & ~3, e4 = e & ~3;
    if (s4 >= e4) {
        for (int j = s + threadIdx.x; j < e; j += RTHREADS) {
            const float2 v0 = rec_val[j];
            fx_add(acc, rec_idx[j], v0.x, v0.y, scale);
        }
    } else {
        if ((int)threadIdx.x < s4 - s) {
            const int j = s + threadIdx.x;
            const float2 v0 = rec_val[j];
            fx_add(acc, rec_idx[j], v0.x, v0.y, scale);
        }
        if ((int)threadIdx.x < e - e4) {
            const int j = e4 + threadIdx.x;
            const float2 v0 = rec_val[j];
            fx_add(acc, rec_idx[j], v0.x, v0.y, scale);
        }
        const ushort4* ri = reinterpret_cast<const ushort4*>(rec_idx);
        const float4* rv = reinterpret_cast<const float4*>(rec_val);
        const int g0 = s4 >> 2, g1 = e4 >> 2;
        int gq = g0 + threadIdx.x;
        for (; gq + RTHREADS < g1; gq += 2 * RTHREADS) {
            const ushort4 ia = ri[gq], ib = ri[gq + RTHREADS];
            const float4 a0 = rv[2 * gq], a1 = rv[2 * gq + 1];
            const float4 b0 = rv[2 * (gq + RTHREADS)], b1 = rv[2 * (gq + RTHREADS) + 1];
            fx_add(acc, ia.x, a0.x, a0.y, scale); fx_add(acc, ia.y, a0.z, a0.w, scale);
            fx_add(acc, ia.z, a1.x, a1.y, scale); fx_add(acc, ia.w, a1.z, a1.w, scale);
            fx_add(acc, ib.x, b0.x, b0.y, scale); fx_add(acc, ib.y, b0.z, b0.w, scale);
            fx_add(acc, ib.z, b1.x, b1.y, scale); fx_add(acc, ib.w, b1.z, b1.w, scale);
        }
        for (; gq < g1; gq += RTHREADS) {
            const ushort4 ia = ri[gq];
            const float4 a0 = rv[2 * gq], a1 = rv[2 * gq + 1];
            fx_add(acc, ia.x, a0.x, a0.y, scale); fx_add(acc, ia.y, a0.z, a0.w, scale);
            fx_add(acc, ia.z, a1.x, a1.y, scale); fx_add(acc, ia.w, a1.z, a1.w, scale);
        }
    }
    __syncthreads();
    float* tab = grad + ((int64_t)cfg.offsets[l] + first) * 2;
    for (int k = threadIdx.x; k < entries; k += RTHREADS) {
        const long long ix = (long long)acc[2 * k], iy = (long long)acc[2 * k + 1];
        if (ix == 0 && iy == 0) continue;
        const float vx = (float)ix * inv_scale, vy = (float)iy * inv_scale;
        if (nparts == 1) {
            float2* t = reinterpret_cast<float2*>(tab) + k;
            float2 o = *t;
            o.x += vx; o.y += vy;
            *t = o;
        } else {
            unsafeAtomicAdd(tab + 2 * k, vx);
            unsafeAtomicAdd(tab + 2 * k + 1, vy);
        }
    }
  }
}

__global__ __launch_bounds__(THREADS) void sh4_kernel(int64_t n, const float* __restrict__ d01, float* __restrict__ out,
                                                       int out_stride)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float x = d01[i * 3 + 0] * 2.f - 1.f, y = d01[i * 3 + 1] * 2.f - 1.f, z = d01[i * 3 + 2] * 2.f - 1.f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float* o = out + i * out_stride;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// d L / d d01 given d L / d sh (x = 2 d01 - 1 => factor 2)
__global__ __launch_bounds__(THREADS) void sh4_bwd_kernel(int64_t n, const float* __restrict__ d01,
                                                           const float* __restrict__ g, int g_stride,
                                                           float* __restrict__ g_d01)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float x = d01[i * 3 + 0] * 2.f - 1.f, y = d01[i * 3 + 1] * 2.f - 1.f, z = d01[i * 3 + 2] * 2.f - 1.f;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    const float* go = g + i * g_stride;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    gy += go[1] * -0.48860251190291987f;
    gz += go[2] * 0.48860251190291987f;
    gx += go[3] * -0.48860251190291987f;
    gx += go[4] * 1.0925484305920792f * y;           gy += go[4] * 1.0925484305920792f * x;
    gy += go[5] * -1.0925484305920792f * z;          gz += go[5] * -1.0925484305920792f * y;
    gz += go[6] * 2.0f * 0.94617469575755997f * z;
    gx += go[7] * -1.0925484305920792f * z;          gz += go[7] * -1.0925484305920792f * x;
    gx += go[8] * 2.0f * 0.54627421529603959f * x;   gy += go[8] * -2.0f * 0.54627421529603959f * y;
    gx += go[9] * 0.59004358992664352f * y * (-6.0f * x);
    gy += go[9] * 0.59004358992664352f * (-3.0f * x2 + 3.0f * y2);
    gx += go[10] * 2.8906114426405538f * y * z;      gy += go[10] * 2.8906114426405538f * x * z;
    gz += go[10] * 2.8906114426405538f * x * y;
    gy += go[11] * 0.45704579946446572f * (1.0f - 5.0f * z2);
    gz += go[11] * 0.45704579946446572f * y * (-10.0f * z);
    gz += go[12] * 0.3731763325901154f * (15.0f * z2 - 3.0f);
    gx += go[13] * 0.45704579946446572f * (1.0f - 5.0f * z2);
    gz += go[13] * 0.45704579946446572f * x * (-10.0f * z);
    gx += go[14] * 1.4453057213202769f * z * 2.0f * x; gy += go[14] * 1.4453057213202769f * z * -2.0f * y;
    gz += go[14] * 1.4453057213202769f * (x2 - y2);
    gx += go[15] * 0.59004358992664352f * (-3.0f * x2 + 3.0f * y2);
    gy += go[15] * 0.59004358992664352f * x * 6.0f * y;
    g_d01[i * 3 + 0] = 2.0f * gx; g_d01[i * 3 + 1] = 2.0f * gy; g_d01[i * 3 + 2] = 2.0f * gz;
}

// contractions with the stored Jacobian J [n, K, 3] (K = n_levels * 2):
//   MODE 0: out[n,3]  = sum_k v[n,k] * J[n,k,:]      (d L / d x   from d L / d enc)
//   MODE 1: out[n,K]  = J[n,k,:] . q[n,:]            (JVP of the encoding along q)
template <int MODE>
__global__ __launch_bounds__(THREADS) void jac_contract_kernel(int64_t n, int K, const float* __restrict__ jac,
                                                                const float* __restrict__ v, int v_stride,
                                                                float* __restrict__ out, int out_stride)
{
    if (MODE == 0) {
        const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        if (i >= n) return;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const float* J = jac + i * K * 3;
        for (int k = 0; k < K; k++) {
            const float g = v[i * v_stride + k];
            a0 = fmaf(g, J[k * 3 + 0], a0); a1 = fmaf(g, J[k * 3 + 1], a1); a2 = fmaf(g, J[k * 3 + 2], a2);
        }
        out[i * out_stride + 0] = a0; out[i * out_stride + 1] = a1; out[i * out_stride + 2] = a2;
    } else {
        const int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        const int64_t i = t / K;
        const int k = (int)(t % K);
        if (i >= n) return;
        const float* J = jac + (i * K + k) * 3;
        out[i * out_stride + k] = J[0] * v[i * v_stride + 0] + J[1] * v[i * v_stride + 1] + J[2] * v[i * v_stride + 2];
    }
}

}  // namespace

IA_EXPORT int ia_hashgrid_jac_contract(int mode, int64_t n, int K, const float* jac, const float* v, int v_stride,
                                       float* out, int out_stride, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(mode == 0 || mode == 1, "mode 0 (J^T v) or 1 (J q)");
    if (mode == 0)
        jac_contract_kernel<0><<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, K, jac, v, v_stride, out, out_stride);
    else
        jac_contract_kernel<1><<<ia::cdiv(n * K, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, K, jac, v, v_stride, out, out_stride);
    return ia::check_launch("ia_hashgrid_jac_contract");
}

IA_EXPORT int ia_sh4_bwd(int64_t n, const float* d01, const float* g_sh, int g_stride, float* g_d01, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    sh4_bwd_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, d01, g_sh, g_stride, g_d01);
    return ia::check_launch("ia_sh4_bwd");
}

IA_EXPORT int64_t ia_hashgrid_n_entries(int n_levels, int log2_hashmap_size, int base_resolution, float per_level_scale)
{
    if (n_levels <= 0 || n_levels > MAX_LEVELS) return -1;
    HashCfg c;
    make_cfg(c, n_levels, log2_hashmap_size, base_resolution, per_level_scale);
    return (int64_t)c.offsets[n_levels];
}

IA_EXPORT int ia_hashgrid_fwd(int64_t n, const float* x, const float* params, int n_levels, int n_features,
                              int log2_hashmap_size, int base_resolution, float per_level_scale, float* out,
                              int out_stride, float* dy_dx, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n_features == 2, "n_features_per_level must be 2 on this path");
    IA_REQUIRE(n_levels > 0 && n_levels <= MAX_LEVELS, "n_levels out of range");
    IA_REQUIRE(out_stride >= n_levels * 2 && (out_stride % 2) == 0, "out_stride must be even and >= n_levels*2");
    HashCfg c;
    make_cfg(c, n_levels, log2_hashmap_size, base_resolution, per_level_scale);
    const int grid = ia::cdiv(n * n_levels, THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (dy_dx) hash_fwd_kernel<true><<<grid, THREADS, 0, s>>>(n, x, (const float2*)params, c, out, out_stride, dy_dx);
    else hash_fwd_kernel<false><<<grid, THREADS, 0, s>>>(n, x, (const float2*)params, c, out, out_stride, nullptr);
    return ia::check_launch("ia_hashgrid_fwd");
}


IA_EXPORT int64_t ia_hashgrid_fwd_scratch_bytes(int64_t n, int n_levels, int with_jac)
{
    if (n < 0 || n_levels <= 0 || n_levels > MAX_LEVELS) return -1;
    return n * n_levels * 8 + (with_jac ? n * n_levels * 24 : 0) + 256;
}

// same outputs as ia_hashgrid_fwd; scratch = ia_hashgrid_fwd_scratch_bytes(n, n_levels, dy_dx != NULL) bytes
IA_EXPORT int ia_hashgrid_fwd_xcd(int64_t n, const float* x, const float* params, int n_levels, int n_features,
                                  int log2_hashmap_size, int base_resolution, float per_level_scale, float* out,
                                  int out_stride, float* dy_dx, void* scratch, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n_features == 2, "n_features_per_level must be 2 on this path");
    IA_REQUIRE(n_levels > 0 && n_levels <= MAX_LEVELS, "n_levels out of range");
    IA_REQUIRE((reinterpret_cast<uintptr_t>(params) & 15) == 0, "ia_hashgrid_fwd_xcd: the table must be 16-byte aligned (entry pairs are read with one aligned 16-byte load)");
    IA_REQUIRE(scratch != nullptr, "scratch required");
    HashCfg c;
    make_cfg(c, n_levels, log2_hashmap_size, base_resolution, per_level_scale);
    float2* tmp = (float2*)scratch;
    float* tmp_jac = dy_dx ? (float*)((char*)scratch + ((n * n_levels * 8 + 255) / 256) * 256) : nullptr;
    // level sets: "big" levels own a full 2^log2_hashmap_size table (one L2 each), the rest are the small dense ones
    const uint32_t cap = 1u << log2_hashmap_size;
    int n_small = 0;
    while (n_small < n_levels && c.offsets[n_small + 1] - c.offsets[n_small] < cap) n_small++;
    hipStream_t s = (hipStream_t)stream;
    // workgroups per XCD slot = what is RESIDENT at once (the kernels are grid-stride loops over equal shares): with 70 VGPRs the
    // gather kernel fits 7 workgroups of 256 threads per CU, and the 8th per CU of a 256-per-slot grid ran as a second round at
    // an eighth of the occupancy -- a tail as long as the first round
    static int chunks_jac = 0, chunks_plain = 0;
    if (!chunks_plain) {
        int dev = 0, n_cu = 256, b0 = 0, b1 = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b0, hash_fwd_xcd_kernel<false>, THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, hash_fwd_xcd_kernel<true>, THREADS, 0);
        (void)hipGetLastError();
        const int per_xcd = n_cu >= 8 ? n_cu / 8 : 1;
        // ... and FEWER resident workgroups are faster still: the resident workgroups of a slot sweep a window of
        // chunks x 512 consecutive (spatially sorted) points, and the smaller that window the more of its gathers hit L1 / L2
        // (100 M sorted points, ms per forward: 256 -> 37.7, 224 -> 33.4, 192 -> 31.4, 160 -> 30.6, 128 -> 30.5, 96 -> 33.7, 64 -> 46.4)
        // headline step (ms per step in this kernel, whole workgroups per CU only -- 144 / 176 leave CUs unevenly loaded: 157 / 162):
        // 4 per CU 149.6, 5 per CU 144.2, 6 per CU 152.6, 8 per CU (7 resident) 186.4
        // (those figures are for the per-XCD-level passes; with one table at a time: 4 per CU 22.9 ms per 100 M points, 5 -> 21.6,
        //  6 -> 21.0, 7 -> 20.9; headline step 116.3 / 113.5 / 113.9 ms for 5 / 6 / 7)
        chunks_plain = (b0 > 6 ? 6 : (b0 > 0 ? b0 : 4)) * per_xcd;
        chunks_jac = (b1 > 6 ? 6 : (b1 > 0 ? b1 : 4)) * per_xcd;
        if (const char* e = getenv("IA_HASH_XCD_CHUNKS")) chunks_plain = chunks_jac = atoi(e);
    }
    const int chunks = dy_dx ? chunks_jac : chunks_plain;
    int l = n_small;
    bool small_done = (n_small == 0);
    // Default schedule: ONE table at a time.  A launch gathers one hashed level (or the whole dense set) for all points, the
    // eight XCD slots taking an eighth of the points each: every XCD's L2 holds just that 4 MB table, and the launches are
    // balanced by construction.  Giving each XCD its own level (passes A / B below, IA_HASH_XCD_PLAN=passes) is bound by the
    // slowest level of a pass: per-level cost for 100 M sorted points on the whole device is 1.18 ms (level 5) ... 1.83
    // (level 12) ... 2.42 ms (level 15), 3.47 ms for the five dense levels together = 21.8 ms against 26.9 ms for the passes.
    // Small batches (< 300 k points: the reference's 4096-ray training batches) take the two passes instead: twelve launches of a batch that
    // fills the device for a few microseconds cost more than the passes' imbalance (tools/small_gather_probe.py, gather ms, level / passes:
    // 140 k points 0.184 / 0.101, 206 k 0.223 / 0.147, 557 k 0.287 / 0.335, 1 M 0.445 / 0.581, 10 M 2.19 / 3.02).
    static const int plan_env = getenv("IA_HASH_XCD_PLAN") ? (getenv("IA_HASH_XCD_PLAN")[0] == 'p' ? 1 : 2) : 0;      // 1: passes, 2: by level, 0: by size
    const bool by_level = plan_env == 2 || (plan_env == 0 && n >= 300000);
    // experiment knob (round 5): hashed levels per launch (default 1).  Two levels per launch read the coordinates half as often and put
    // 8 MB of tables in front of every XCD's 4 MB L2 -- measured: DESIGN 4.3.
    static const int lpl = getenv("IA_HASH_LEVELS_PER_LAUNCH") ? max(1, atoi(getenv("IA_HASH_LEVELS_PER_LAUNCH"))) : 1;
    if (by_level) {
        for (int u = (n_small > 0 ? -1 : 0); u < n_levels - n_small; u += (u < 0 ? 1 : lpl)) {
            XcdPlan plan;
            plan.straight = 1;
            for (int k = 0; k < 8; k++) {
                plan.first_level[k] = u < 0 ? 0 : n_small + u;
                plan.n_level[k] = u < 0 ? n_small : min(lpl, n_levels - n_small - u);
                plan.part[k] = k; plan.nparts[k] = 8;
            }
            if (dy_dx) hash_fwd_xcd_kernel<true><<<8 * chunks, THREADS, 0, s>>>(n, x, (const float2*)params, c, plan, tmp, tmp_jac);
            else hash_fwd_xcd_kernel<false><<<8 * chunks, THREADS, 0, s>>>(n, x, (const float2*)params, c, plan, tmp, nullptr);
        }
        l = n_levels;
        small_done = true;
    }
    while (l < n_levels || !small_done) {
        XcdPlan plan;
        plan.straight = 1;
        const int big_left = n_levels - l;
        if (big_left >= 8) {                                    // pass A: one big level per XCD, all points
            for (int k = 0; k < 8; k++) { plan.first_level[k] = l + k; plan.n_level[k] = 1; plan.part[k] = 0; plan.nparts[k] = 1; }
            l += 8;
        } else {                                                // pass B: remaining big levels + the small set
            // XCD slots by COST of each unit, every unit at least one slot; a unit with several slots splits the points between
            // them.  Cost of one hashed level = 1 (every gather of a fine level misses L1); the dense levels hit L1 / L2 on spatially
            // sorted points, so the whole small set costs about as much as ONE hashed level -- weighing it by its level count (5)
            // left three XCDs with a full hashed level each while five XCDs idled through a fifth of the small set.
            const int units = big_left + (small_done ? 0 : 1);
            static const float small_cost = getenv("IA_HASH_SMALL_COST") ? (float)atof(getenv("IA_HASH_SMALL_COST")) : 1.0f;
            float w[9];
            int share[9], used = 0;
            for (int u = 0; u < units; u++) { w[u] = (!small_done && u == units - 1) ? small_cost : 1.0f; share[u] = 1; used++; }
            for (; used < 8; used++) {                          // next slot to the unit with the most work per slot
                int best = 0;
                for (int u = 1; u < units; u++) if (w[u] / share[u] > w[best] / share[best]) best = u;
                share[best]++;
            }
            int k = 0;
            for (int u = 0; u < units; u++) {
                const bool is_small = (!small_done && u == units - 1);
                for (int j = 0; j < share[u]; j++, k++) {
                    plan.first_level[k] = is_small ? 0 : l + u;
                    plan.n_level[k] = is_small ? n_small : 1;
                    plan.part[k] = j; plan.nparts[k] = share[u];
                }
            }
            l = n_levels;
            small_done = true;
        }
        if (dy_dx) hash_fwd_xcd_kernel<true><<<8 * chunks, THREADS, 0, s>>>(n, x, (const float2*)params, c, plan, tmp, tmp_jac);
        else hash_fwd_xcd_kernel<false><<<8 * chunks, THREADS, 0, s>>>(n, x, (const float2*)params, c, plan, tmp, nullptr);
    }
    if (out == nullptr) return ia::check_launch("ia_hashgrid_fwd_xcd");      // level-major result stays in `scratch` (float2 [L][n])
    const size_t lds = sizeof(float) * THREADS * (dy_dx ? 6 * n_levels + 1 : 2 * n_levels + 1);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)hash_transpose_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipGetLastError();
        attr = true;
    }
    if (dy_dx) hash_transpose_kernel<true><<<ia::cdiv(n, THREADS), THREADS, lds, s>>>(n, n_levels, tmp, tmp_jac, out, out_stride, dy_dx);
    else hash_transpose_kernel<false><<<ia::cdiv(n, THREADS), THREADS, lds, s>>>(n, n_levels, tmp, nullptr, out, out_stride, nullptr);
    return ia::check_launch("ia_hashgrid_fwd_xcd");
}

// level-major results only: float2 [L][n] at scratch, and (with_jac) float [L][n][6] at scratch + ia_hashgrid_fwd_levels_jac_offset
IA_EXPORT int64_t ia_hashgrid_fwd_levels_jac_offset(int64_t n, int n_levels)
{
    return ((n * n_levels * 8 + 255) / 256) * 256;
}

IA_EXPORT int ia_hashgrid_fwd_levels(int64_t n, const float* x, const float* params, int n_levels, int n_features, int log2_hashmap_size,
                                     int base_resolution, float per_level_scale, int with_jac, void* scratch, ia_stream_t stream)
{
    // out == NULL: no transpose; a non-NULL dy_dx only selects the Jacobian variant of the gather (nothing is written through it)
    return ia_hashgrid_fwd_xcd(n, x, params, n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale, nullptr, 0,
                               with_jac ? reinterpret_cast<float*>(scratch) : nullptr, scratch, stream);
}

IA_EXPORT int ia_hashgrid_bwd(int64_t n, const float* x, int n_levels, int n_features, int log2_hashmap_size,
                              int base_resolution, float per_level_scale, const float* g_enc, int g_enc_stride,
                              const float* g_jac, int g_jac_stride, const float* q, float* grad_params,
                              ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n_features == 2, "n_features_per_level must be 2 on this path");
    IA_REQUIRE(n_levels > 0 && n_levels <= MAX_LEVELS, "n_levels out of range");
    IA_REQUIRE(g_enc != nullptr || g_jac != nullptr, "need g_enc and/or g_jac");
    IA_REQUIRE((g_jac == nullptr) == (q == nullptr), "g_jac and q go together");
    HashCfg c;
    make_cfg(c, n_levels, log2_hashmap_size, base_resolution, per_level_scale);
    hipStream_t s = (hipStream_t)stream;
    const int grid = ia::cdiv(n, THREADS);     // run-merged kernel: one lane per point
    if (g_jac) hash_bwd_runs_kernel<true><<<grid, THREADS, 0, s>>>(n, x, c, g_enc, g_enc_stride, g_jac, g_jac_stride, q, grad_params);
    else hash_bwd_runs_kernel<false><<<grid, THREADS, 0, s>>>(n, x, c, g_enc, g_enc_stride, nullptr, 0, nullptr, grad_params);
    return ia::check_launch("ia_hashgrid_bwd");
}


namespace {
struct BinLayout {
    int nwg; int64_t m, records;
    int64_t off_counts, off_total, off_tmp, off_val, off_idx, bytes;
};
bool staged_fill()
{
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("IA_HASHBWD_FILL");        // test hook: "direct" = per-workgroup runs, scattered appends
        mode = (e && e[0] == 'd') ? 0 : 1;
    }
    return mode == 1;
}

BinLayout bin_layout(int64_t n, int n_levels, int n_buckets)
{
    BinLayout L;
    L.nwg = staged_fill() ? ia::cdiv(n, U_PTS) : ia::cdiv(n, BIN_TILE);       // units of the count matrix
    L.m = (int64_t)n_buckets * L.nwg;
    L.records = n * n_levels * 8;
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    L.off_counts = 0;
    L.off_total = up(L.m * 4);
    L.off_tmp = L.off_total + 512 + 4 * (MAX_BUCKETS + 64);      // total (4 B), level_max[32] at +256, plan at +512
    L.off_val = up(L.off_tmp + ia_scan_tmp_bytes(L.m));
    L.off_idx = up(L.off_val + L.records * 8);
    L.bytes = up(L.off_idx + L.records * 2);
    return L;
}
}  // namespace

IA_EXPORT int64_t ia_hashgrid_bwd_scratch_bytes(int64_t n, int n_levels, int log2_hashmap_size, int base_resolution,
                                                float per_level_scale)
{
    if (n_levels <= 0 || n_levels > MAX_LEVELS || n < 0) return -1;
    HashCfg c;
    make_cfg(c, n_levels, log2_hashmap_size, base_resolution, per_level_scale);
    BinCfg b;
    make_bins(b, c);
    if (b.n_buckets > MAX_BUCKETS) return -1;
    return bin_layout(n, n_levels, b.n_buckets).bytes;
}

IA_EXPORT int ia_hashgrid_bwd_binned(int64_t n, const float* x, int n_levels, int n_features, int log2_hashmap_size,
                                     int base_resolution, float per_level_scale, const float* g_enc, int g_enc_stride,
                                     const float* g_jac, int g_jac_stride, const float* q, float* grad_params,
                                     uint32_t level_mask, void* scratch, int64_t scratch_bytes, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n_features == 2, "n_features_per_level must be 2 on this path");
    IA_REQUIRE(n_levels > 0 && n_levels <= MAX_LEVELS, "n_levels out of range");
    IA_REQUIRE(g_enc != nullptr || g_jac != nullptr, "need g_enc and/or g_jac");
    IA_REQUIRE((g_jac == nullptr) == (q == nullptr), "g_jac and q go together");
    IA_REQUIRE(n * n_levels * 8 < (int64_t)0x7FFFFFFF, "n too large for 32-bit record offsets: split the batch");
    HashCfg c;
    make_cfg(c, n_levels, log2_hashmap_size, base_resolution, per_level_scale);
    BinCfg b;
    make_bins(b, c);
    IA_REQUIRE(b.n_buckets <= MAX_BUCKETS, "too many (level, slice) buckets");
    const BinLayout L = bin_layout(n, n_levels, b.n_buckets);
    IA_REQUIRE(scratch != nullptr && scratch_bytes >= L.bytes, "scratch smaller than ia_hashgrid_bwd_scratch_bytes(n)");
    char* base = (char*)scratch;
    int32_t* counts = (int32_t*)(base + L.off_counts);
    int32_t* total = (int32_t*)(base + L.off_total);
    void* tmp = base + L.off_tmp;
    float2* rec_val = (float2*)(base + L.off_val);
    uint16_t* rec_idx = (uint16_t*)(base + L.off_idx);
    hipStream_t s = (hipStream_t)stream;
    unsigned* level_max = (unsigned*)(base + L.off_total + 256);
    if (hipMemsetAsync(level_max, 0, MAX_LEVELS * sizeof(unsigned), s) != hipSuccess) return ia::check_launch("ia_hashgrid_bwd_binned(memset)");
    const bool staged = staged_fill();
    const int ugrid = ia::cdiv(L.nwg, U_WAVES);
    constexpr size_t lds_count = (size_t)U_WAVES * MAX_BUCKETS * sizeof(int), lds_fill = (size_t)U_WAVES * sizeof(UnitLds);
    if (staged) {
        if (g_jac) hash_bin_unit_kernel<true, false><<<ugrid, THREADS, lds_count, s>>>(n, x, c, b, g_enc, g_enc_stride, g_jac, g_jac_stride, q, L.nwg, L.m, counts, total, rec_idx, rec_val, level_mask, level_max);
        else hash_bin_unit_kernel<false, false><<<ugrid, THREADS, lds_count, s>>>(n, x, c, b, g_enc, g_enc_stride, nullptr, 0, nullptr, L.nwg, L.m, counts, total, rec_idx, rec_val, level_mask, level_max);
    } else {
        if (g_jac) hash_bin_kernel<true, false><<<L.nwg, THREADS, 0, s>>>(n, x, c, b, g_enc, g_enc_stride, g_jac, g_jac_stride, q, L.nwg, counts, rec_idx, rec_val, level_mask, level_max);
        else hash_bin_kernel<false, false><<<L.nwg, THREADS, 0, s>>>(n, x, c, b, g_enc, g_enc_stride, nullptr, 0, nullptr, L.nwg, counts, rec_idx, rec_val, level_mask, level_max);
    }
    int r = ia_exclusive_scan_i32(counts, counts, total, L.m, tmp, stream);
    if (r != IA_OK) return r;
    static bool attr_fill = false;
    if (staged && !attr_fill) {
        (void)hipFuncSetAttribute((const void*)hash_bin_unit_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill);
        (void)hipFuncSetAttribute((const void*)hash_bin_unit_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill);
        (void)hipGetLastError();
        attr_fill = true;
    }
    if (staged) {
        if (g_jac) hash_bin_unit_kernel<true, true><<<ugrid, THREADS, lds_fill, s>>>(n, x, c, b, g_enc, g_enc_stride, g_jac, g_jac_stride, q, L.nwg, L.m, counts, total, rec_idx, rec_val, level_mask, level_max);
        else hash_bin_unit_kernel<false, true><<<ugrid, THREADS, lds_fill, s>>>(n, x, c, b, g_enc, g_enc_stride, nullptr, 0, nullptr, L.nwg, L.m, counts, total, rec_idx, rec_val, level_mask, level_max);
    } else {
        if (g_jac) hash_bin_kernel<true, true><<<L.nwg, THREADS, 0, s>>>(n, x, c, b, g_enc, g_enc_stride, g_jac, g_jac_stride, q, L.nwg, counts, rec_idx, rec_val, level_mask, level_max);
        else hash_bin_kernel<false, true><<<L.nwg, THREADS, 0, s>>>(n, x, c, b, g_enc, g_enc_stride, nullptr, 0, nullptr, L.nwg, counts, rec_idx, rec_val, level_mask, level_max);
    }
    static bool attr = false;
    constexpr size_t red_lds = (size_t)SLICE * 2 * sizeof(unsigned long long);      // 128 KB
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)hash_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)red_lds);
        (void)hipGetLastError();
        attr = true;
    }
    int32_t* plan = (int32_t*)(base + L.off_total + 512);
    hash_plan_kernel<<<1, 1024, 0, s>>>(b, L.nwg, counts, total, plan);
    hash_reduce_kernel<<<256, RTHREADS, red_lds, s>>>(c, b, L.nwg, counts, total, rec_idx, rec_val, level_max, plan, grad_params);
    return ia::check_launch("ia_hashgrid_bwd_binned");
}

IA_EXPORT int ia_sh4_fwd(int64_t n, const float* d01, float* out, int out_stride, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(out_stride >= 16, "out_stride must be >= 16");
    sh4_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, d01, out, out_stride);
    return ia::check_launch("ia_sh4_fwd");
}
