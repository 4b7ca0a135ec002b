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
    return index % hsize;
}

// forward (+ optional analytic d enc / d x)
template <bool WITH_JAC>
__global__ __launch_bounds__(THREADS) void hash_fwd_kernel(int64_t n, const float* __restrict__ x,
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
        a0 += w * v[c].x;
        a1 += w * v[c].y;
        if (WITH_JAC) {
            const float dx = ((c & 1) ? sc : -sc) * wy * wz;
            const float dy = ((c & 2) ? sc : -sc) * wx * wz;
            const float dz = ((c & 4) ? sc : -sc) * wx * wy;
            j0[0] += dx * v[c].x; j0[1] += dy * v[c].x; j0[2] += dz * v[c].x;
            j1[0] += dx * v[c].y; j1[1] += dy * v[c].y; j1[2] += dz * v[c].y;
        }
    }
    *reinterpret_cast<float2*>(out + i * out_stride + l * 2) = make_float2(a0, a1);
    if (WITH_JAC) {
        float* J = dy_dx + (i * L * 2 + l * 2) * 3;
        J[0] = j0[0]; J[1] = j0[1]; J[2] = j0[2];
        J[3] = j1[0]; J[4] = j1[1]; J[5] = j1[2];
    }
}

// backward w.r.t. the table:
//   grad[c] += gE[l,:] * w_c  +  gG[l,:] * sum_a q[a] * d w_c / d x_a      (second term optional)
template <bool SECOND>
__global__ __launch_bounds__(THREADS) void hash_bwd_kernel(int64_t n, const float* __restrict__ x, HashCfg cfg,
                                                            const float* __restrict__ gE, int gE_stride,
                                                            const float* __restrict__ gG, int gG_stride,
                                                            const float* __restrict__ q, float* __restrict__ grad)
{
    const int L = cfg.n_levels;
    const int64_t t = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t i = t / L;
    const int l = (int)(t % L);
    if (i >= n) return;
    const float sc = cfg.scale[l];
    const uint32_t res = cfg.res[l], hsize = cfg.offsets[l + 1] - cfg.offsets[l];
    float* tab = grad + (int64_t)cfg.offsets[l] * 2;
    float pos[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float p = fmaf(sc, x[i * 3 + d], 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)(int)fl;
        pos[d] = p - fl;
    }
    const float2 e = gE ? *reinterpret_cast<const float2*>(gE + i * gE_stride + l * 2) : make_float2(0.f, 0.f);
    float2 g = make_float2(0.f, 0.f);
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (SECOND) {
        g = *reinterpret_cast<const float2*>(gG + i * gG_stride + l * 2);
        qx = q[i * 3 + 0]; qy = q[i * 3 + 1]; qz = q[i * 3 + 2];
    }
    if (e.x == 0.f && e.y == 0.f && g.x == 0.f && g.y == 0.f) return;   // masked-out levels (progressive bands)
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float wx = (c & 1) ? pos[0] : 1.0f - pos[0];
        const float wy = (c & 2) ? pos[1] : 1.0f - pos[1];
        const float wz = (c & 4) ? pos[2] : 1.0f - pos[2];
        float w0 = wx * wy * wz;
        float vx = e.x * w0, vy = e.y * w0;
        if (SECOND) {
            const float dw = ((c & 1) ? sc : -sc) * wy * wz * qx + ((c & 2) ? sc : -sc) * wx * wz * qy +
                             ((c & 4) ? sc : -sc) * wx * wy * qz;
            vx += g.x * dw;
            vy += g.y * dw;
        }
        const uint32_t idx = grid_index(hsize, res, pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1));
        unsafeAtomicAdd(tab + (int64_t)idx * 2 + 0, vx);
        unsafeAtomicAdd(tab + (int64_t)idx * 2 + 1, vy);
    }
}

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

IA_EXPORT int ia_sh4_fwd(int64_t n, const float* d01, float* out, int out_stride, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(out_stride >= 16, "out_stride must be >= 16");
    sh4_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, d01, out, out_stride);
    return ia::check_launch("ia_sh4_fwd");
}
