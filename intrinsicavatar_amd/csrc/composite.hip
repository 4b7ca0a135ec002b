// composite.hip -- transmittance / weight / accumulation along packed rays for gfx950.
// Replaces nerfacc.render_weight_from_alpha and nerfacc.accumulate_along_rays
// (call sites models/intrinsic_avatar.py:506,1199,1427-1453; models/volrend.py:162,176-187,764,783-797)
// and fuses them for the render path (ia_composite_*), see DESIGN.md.
//
// Mapping: one ray per lane walks its own contiguous run of samples (rays are independent,
// the product/sum order is left-to-right, identical to oracle/ia_oracle.c => bit-exact).
// Built with -ffp-contract=off.
#include "ia_common.h"

namespace {

constexpr int THREADS = 256;

// Per-ray serial recurrences (the multiplication / summation order per ray is part of the bit-exact contract), made
// coalesced: a workgroup's 256 rays own ONE contiguous range of samples (packed layout), which is staged through LDS in
// chunks -- cooperative 64-lane-wide loads and stores -- while every lane walks its own ray inside the chunk.  With one
// lane per ray reading global memory directly, each 4-byte access touched its own 64-byte sector (measured 2.5 GB
// fetched for 4.4 M samples in the backward).  If the rays of a workgroup are not contiguous (hand-made packed_info),
// the lanes fall back to direct global accesses.
struct WgSpan { int lo, hi; bool contiguous; };

__device__ __forceinline__ WgSpan wg_span(int2 pi, int* s_red /*[3]*/)
{
    if (threadIdx.x == 0) { s_red[0] = 0x7fffffff; s_red[1] = 0; s_red[2] = 0; }
    __syncthreads();
    if (pi.y > 0) {
        atomicMin(&s_red[0], pi.x);
        atomicMax(&s_red[1], pi.x + pi.y);
        atomicAdd(&s_red[2], pi.y);
    }
    __syncthreads();
    WgSpan w;
    w.lo = s_red[0]; w.hi = s_red[1];
    w.contiguous = (s_red[2] > 0) && (w.hi - w.lo == s_red[2]);
    if (s_red[2] == 0) { w.lo = 0; w.hi = 0; w.contiguous = true; }
    return w;
}

constexpr int CH_FWD = 4096, CH_BWD = 2048, CH_ACC = 2048;

__global__ __launch_bounds__(THREADS) void weight_from_alpha_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                                     const float* __restrict__ alphas,
                                                                     float* __restrict__ weights, float* __restrict__ trans)
{
    __shared__ float sA[CH_FWD], sT[CH_FWD], sW[CH_FWD];
    __shared__ int s_red[3];
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    int2 pi = make_int2(0, 0);
    if (r < n_rays) pi = reinterpret_cast<const int2*>(packed_info)[r];
    const WgSpan w = wg_span(pi, s_red);
    float T = 1.0f;
    if (!w.contiguous) {
        for (int j = 0; j < pi.y; j++) {
            const float a = alphas[pi.x + j];
            trans[pi.x + j] = T;
            weights[pi.x + j] = T * a;
            T = T * (1.0f - a);
        }
        return;
    }
    int pos = pi.x;
    const int end = pi.x + pi.y;
    for (int c0 = w.lo; c0 < w.hi; c0 += CH_FWD) {
        const int c1 = (c0 + CH_FWD < w.hi) ? c0 + CH_FWD : w.hi;
        for (int i = c0 + threadIdx.x; i < c1; i += THREADS) sA[i - c0] = alphas[i];
        __syncthreads();
        while (pos < end && pos < c1) {
            const float a = sA[pos - c0];
            sT[pos - c0] = T;
            sW[pos - c0] = T * a;
            T = T * (1.0f - a);
            pos++;
        }
        __syncthreads();
        for (int i = c0 + threadIdx.x; i < c1; i += THREADS) { trans[i] = sT[i - c0]; weights[i] = sW[i - c0]; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(THREADS) void weight_from_alpha_bwd_kernel(
    int64_t n_rays, const int32_t* __restrict__ packed_info, const float* __restrict__ alphas,
    const float* __restrict__ weights, const float* __restrict__ trans, const float* __restrict__ g_weights,
    const float* __restrict__ g_trans, float* __restrict__ g_alphas)
{
    __shared__ float sA[CH_BWD], sW[CH_BWD], sT[CH_BWD], sGw[CH_BWD], sGt[CH_BWD], sO[CH_BWD];
    __shared__ int s_red[3];
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    int2 pi = make_int2(0, 0);
    if (r < n_rays) pi = reinterpret_cast<const int2*>(packed_info)[r];
    const WgSpan w = wg_span(pi, s_red);
    float S = 0.0f;
    if (!w.contiguous) {
        for (int j = pi.y - 1; j >= 0; j--) {
            const int i = pi.x + j;
            const float gw = g_weights ? g_weights[i] : 0.0f, gT = g_trans ? g_trans[i] : 0.0f;
            g_alphas[i] = gw * trans[i] - S / fmaxf(1.0f - alphas[i], 1e-10f);
            S = S + (gw * weights[i] + gT * trans[i]);
        }
        return;
    }
    int pos = pi.x + pi.y - 1;                     // walks down to pi.x
    for (int c1 = w.hi; c1 > w.lo; c1 -= CH_BWD) {
        const int c0 = (c1 - CH_BWD > w.lo) ? c1 - CH_BWD : w.lo;
        for (int i = c0 + threadIdx.x; i < c1; i += THREADS) {
            sA[i - c0] = alphas[i]; sW[i - c0] = weights[i]; sT[i - c0] = trans[i];
            sGw[i - c0] = g_weights ? g_weights[i] : 0.0f;
            sGt[i - c0] = g_trans ? g_trans[i] : 0.0f;
        }
        __syncthreads();
        while (pi.y > 0 && pos >= pi.x && pos >= c0) {
            const int k = pos - c0;
            const float gw = sGw[k], gT = sGt[k];
            sO[k] = gw * sT[k] - S / fmaxf(1.0f - sA[k], 1e-10f);
            S = S + (gw * sW[k] + gT * sT[k]);
            pos--;
        }
        __syncthreads();
        for (int i = c0 + threadIdx.x; i < c1; i += THREADS) g_alphas[i] = sO[i - c0];
        __syncthreads();
    }
}

// out[r] = sum_j w_j v_j (or sum_j w_j), staged like the kernels above
template <int DIM>
__global__ __launch_bounds__(THREADS) void accumulate_staged_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                                     const float* __restrict__ weights,
                                                                     const float* __restrict__ values, float* __restrict__ out)
{
    __shared__ float sW[CH_ACC], sV[CH_ACC * DIM];
    __shared__ int s_red[3];
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    int2 pi = make_int2(0, 0);
    if (r < n_rays) pi = reinterpret_cast<const int2*>(packed_info)[r];
    const WgSpan w = wg_span(pi, s_red);
    float acc[DIM];
#pragma unroll
    for (int k = 0; k < DIM; k++) acc[k] = 0.0f;
    if (!w.contiguous) {
        for (int j = 0; j < pi.y; j++) {
            const float wj = weights[pi.x + j];
#pragma unroll
            for (int k = 0; k < DIM; k++) acc[k] = acc[k] + (values ? wj * values[(int64_t)(pi.x + j) * DIM + k] : wj);
        }
    } else {
        int pos = pi.x;
        const int end = pi.x + pi.y;
        for (int c0 = w.lo; c0 < w.hi; c0 += CH_ACC) {
            const int c1 = (c0 + CH_ACC < w.hi) ? c0 + CH_ACC : w.hi;
            for (int i = c0 + threadIdx.x; i < c1; i += THREADS) sW[i - c0] = weights[i];
            if (values)
                for (int i = c0 * DIM + threadIdx.x; i < c1 * DIM; i += THREADS) sV[i - c0 * DIM] = values[i];
            __syncthreads();
            while (pos < end && pos < c1) {
                const float wj = sW[pos - c0];
#pragma unroll
                for (int k = 0; k < DIM; k++) acc[k] = acc[k] + (values ? wj * sV[(pos - c0) * DIM + k] : wj);
                pos++;
            }
            __syncthreads();
        }
    }
    if (r < n_rays) {
#pragma unroll
        for (int k = 0; k < DIM; k++) out[r * DIM + k] = acc[k];
    }
}

template <int DIM>
__global__ __launch_bounds__(THREADS) void accumulate_kernel(int64_t n_rays, const int32_t* __restrict__ packed_info,
                                                              int dim_rt, const float* __restrict__ weights,
                                                              const float* __restrict__ values, float* __restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const int2 pi = reinterpret_cast<const int2*>(packed_info)[r];
    if (DIM > 0) {
        float acc[DIM > 0 ? DIM : 1];
#pragma unroll
        for (int k = 0; k < DIM; k++) acc[k] = 0.0f;
        for (int j = 0; j < pi.y; j++) {
            const float w = weights[pi.x + j];
#pragma unroll
            for (int k = 0; k < DIM; k++)
                acc[k] = acc[k] + (values ? w * values[(int64_t)(pi.x + j) * DIM + k] : w);
        }
#pragma unroll
        for (int k = 0; k < DIM; k++) out[r * DIM + k] = acc[k];
    } else {
        for (int k = 0; k < dim_rt; k++) {
            float acc = 0.0f;
            for (int j = 0; j < pi.y; j++)
                acc = acc + weights[pi.x + j] * values[(int64_t)(pi.x + j) * dim_rt + k];
            out[r * dim_rt + k] = acc;
        }
    }
}

__global__ __launch_bounds__(THREADS) void accumulate_bwd_kernel(int64_t n_samples, int dim,
                                                                  const int64_t* __restrict__ ray_indices,
                                                                  const float* __restrict__ weights,
                                                                  const float* __restrict__ values,
                                                                  const float* __restrict__ g_out,
                                                                  float* __restrict__ g_weights,
                                                                  float* __restrict__ g_values)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n_samples) return;
    const int64_t r = ray_indices[i];
    const float w = weights[i];
    float gw = 0.0f;
    for (int k = 0; k < dim; k++) {
        const float g = g_out[r * dim + k];
        if (values) gw = gw + g * values[i * dim + k];
        else gw = gw + g;
        if (g_values) g_values[i * dim + k] = w * g;
    }
    if (g_weights) g_weights[i] = gw;
}

}  // namespace

IA_EXPORT int ia_render_weight_from_alpha(int64_t n_rays, const int32_t* packed_info, const float* alphas,
                                          float* weights, float* trans, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    weight_from_alpha_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, packed_info, alphas,
                                                                                             weights, trans);
    return ia::check_launch("ia_render_weight_from_alpha");
}

IA_EXPORT int ia_render_weight_from_alpha_bwd(int64_t n_rays, const int32_t* packed_info, const float* alphas,
                                              const float* weights, const float* trans, const float* g_weights,
                                              const float* g_trans, float* g_alphas, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    weight_from_alpha_bwd_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_rays, packed_info, alphas, weights, trans, g_weights, g_trans, g_alphas);
    return ia::check_launch("ia_render_weight_from_alpha_bwd");
}

IA_EXPORT int ia_accumulate_along_rays(int64_t n_rays, const int32_t* packed_info, int dim, const float* weights,
                                       const float* values, float* out, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(dim >= 1, "dim must be >= 1");
    IA_REQUIRE(values != nullptr || dim == 1, "values == NULL requires dim == 1");
    const int grid = ia::cdiv(n_rays, THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (dim == 1) accumulate_staged_kernel<1><<<grid, THREADS, 0, s>>>(n_rays, packed_info, weights, values, out);
    else if (dim == 3) accumulate_staged_kernel<3><<<grid, THREADS, 0, s>>>(n_rays, packed_info, weights, values, out);
    else accumulate_kernel<0><<<grid, THREADS, 0, s>>>(n_rays, packed_info, dim, weights, values, out);
    return ia::check_launch("ia_accumulate_along_rays");
}

IA_EXPORT int ia_accumulate_along_rays_bwd(int64_t n_samples, int dim, const int64_t* ray_indices,
                                           const float* weights, const float* values, const float* g_out,
                                           float* g_weights, float* g_values, ia_stream_t stream)
{
    if (n_samples == 0) return IA_OK;
    accumulate_bwd_kernel<<<ia::cdiv(n_samples, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_samples, dim, ray_indices, weights, values, g_out, g_weights, g_values);
    return ia::check_launch("ia_accumulate_along_rays_bwd");
}
