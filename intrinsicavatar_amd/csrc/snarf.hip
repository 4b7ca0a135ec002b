// snarf.hip -- fast-SNARF deformer kernels for gfx950.
// Replaces
//   K10 precompute_kernel  models/deformers/fast_snarf/cuda/precompute/precompute.cu:24-103
//   K8  broyden_kernel     models/deformers/fast_snarf/cuda/fuse_kernel/fuse_cuda_kernel_fast.cu:250-452
//   K9  filter             models/deformers/fast_snarf/cuda/filter/filter.cu:10-77
//
// MI355X layout decision: the 12-channel voxel_J grid is kept CHANNEL-LAST
// ([B,D,H,W,12], 48 B per voxel) next to the reference's [B,12,D,H,W].  A trilinear fetch then
// touches 4 x 96 contiguous bytes (x0,x1 pairs) = 24 dwordx4 loads, instead of 96 scattered
// dwords on 96 different cache lines; the 25 MB grid lives in the 256 MiB Infinity Cache.
// No device synchronisation after the launches (the reference calls cudaDeviceSynchronize()).
//
// Arithmetic order is identical to oracle/ia_oracle.c (TU built with -ffp-contract=off), so
// is_valid / filter masks are bit-exact and x, J_inv match bit-for-bit.
#include <stdlib.h>

#include "ia_common.h"
#include "ia_zero.h"

namespace {

constexpr int THREADS = 256;

// ---- K10 ------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void precompute_kernel(int B, int D, int H, int W,
                                                              const float* __restrict__ voxel_w,
                                                              const float* __restrict__ tfs,
                                                              const float* __restrict__ offset,
                                                              const float* __restrict__ scale,
                                                              float* __restrict__ voxel_d, float* __restrict__ voxel_J,
                                                              float* __restrict__ voxel_J_cl)
{
    __shared__ float s_tfs[24 * 12];
    const int64_t vol = (int64_t)D * H * W;
    const int idx_b = blockIdx.y;
    for (int t = threadIdx.x; t < 24 * 12; t += THREADS) {
        const int j = t / 12, c = t % 12;
        s_tfs[t] = tfs[((int64_t)idx_b * 24 + j) * 16 + c];   // rows 0..2 of the 4x4 = first 12 floats
    }
    __syncthreads();
    const int64_t v = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (v >= vol) return;
    const int idx_d = (int)(v / ((int64_t)H * W));
    const int idx_h = (int)(v % ((int64_t)H * W) / W);
    const int idx_w = (int)(v % ((int64_t)H * W) % W);
    const float coord_x = (((float)idx_w) / (W - 1) * 2 - 1) / scale[0] - offset[0];
    const float coord_y = (((float)idx_h) / (H - 1) * 2 - 1) / scale[1] - offset[1];
    const float coord_z = (((float)idx_d) / (D - 1) * 2 - 1) / scale[2] - offset[2];
    float J[12];
#pragma unroll
    for (int c = 0; c < 12; c++) J[c] = 0;
    for (int j = 0; j < 24; j++) {
        const float wj = voxel_w[j * vol + v];   // coalesced across lanes
#pragma unroll
        for (int c = 0; c < 12; c++) J[c] += wj * s_tfs[j * 12 + c];
    }
    if (voxel_J) {
#pragma unroll
        for (int c = 0; c < 12; c++) voxel_J[((int64_t)idx_b * 12 + c) * vol + v] = J[c];
    }
    if (voxel_J_cl) {
        float4* dst = reinterpret_cast<float4*>(voxel_J_cl + ((int64_t)idx_b * vol + v) * 12);
        dst[0] = make_float4(J[0], J[1], J[2], J[3]);
        dst[1] = make_float4(J[4], J[5], J[6], J[7]);
        dst[2] = make_float4(J[8], J[9], J[10], J[11]);
    }
    if (voxel_d) {
#pragma unroll
        for (int i0 = 0; i0 < 3; i0++) {
            const float xi = J[i0 * 4 + 0] * coord_x + J[i0 * 4 + 1] * coord_y + J[i0 * 4 + 2] * coord_z + J[i0 * 4 + 3];
            voxel_d[((int64_t)idx_b * 3 + i0) * vol + v] = xi;
        }
    }
}

// ---- trilinear 12-channel fetch -------------------------------------------------
template <int LAYOUT>
__device__ __forceinline__ void load_corner(const float* __restrict__ vJ, int64_t vol, int64_t lin, float c[12])
{
    if (LAYOUT == IA_LAYOUT_NDHWC) {
        // vJ is the kernel-uniform grid base and `lin` a 32-bit voxel index (batch offset included): the loads use the
        // scalar-base + 32-bit vector-offset addressing mode instead of a 64-bit multiply-add per corner
        const char* base = reinterpret_cast<const char*>(vJ);
        const uint32_t boff = (uint32_t)lin * 48u;                 // byte offset < 2^32 (checked at the entry point)
        const float4 a = *reinterpret_cast<const float4*>(base + boff);
        const float4 b = *reinterpret_cast<const float4*>(base + boff + 16u);
        const float4 d = *reinterpret_cast<const float4*>(base + boff + 32u);
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
        c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
        c[8] = d.x; c[9] = d.y; c[10] = d.z; c[11] = d.w;
    } else {
#pragma unroll
        for (int k = 0; k < 12; k++) c[k] = vJ[k * vol + lin];
    }
}

template <int LAYOUT, bool PAIRS = false>
__device__ __forceinline__ void grid_sample_J(const float* __restrict__ vJ, int vox0, int D, int H, int W, float gx, float gy,
                                              float gz, float out[12])
{
    const int64_t vol = (int64_t)D * H * W;
    float ix = ((gx + 1.f) / 2) * (W - 1);
    float iy = ((gy + 1.f) / 2) * (H - 1);
    float iz = ((gz + 1.f) / 2) * (D - 1);
    // (2147483646.0f IS 2^31 in fp32: "ix > 2147483646.0f || ix < -2147483648.0f || !isfinite(ix)" == "not |ix| <= 2^31")
    if (!(fabsf(ix) <= 2147483648.0f)) ix = -100.0f;
    if (!(fabsf(iy) <= 2147483648.0f)) iy = -100.0f;
    if (!(fabsf(iz) <= 2147483648.0f)) iz = -100.0f;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    // (float)x0 == fx and (float)x1 == fx + 1 wherever a corner can be inside the grid (|ix| < 2^24); beyond that no corner
    // is in range and no weight is used
    const float ax1 = (fx + 1.0f) - ix, ax0 = ix - fx, ay1 = (fy + 1.0f) - iy, ay0 = iy - fy, az1 = (fz + 1.0f) - iz, az0 = iz - fz;
    const float wgt[8] = {
        ax1 * ay1 * az1, ax0 * ay1 * az1,
        ax1 * ay0 * az1, ax0 * ay0 * az1,
        ax1 * ay1 * az0, ax0 * ay1 * az0,
        ax1 * ay0 * az0, ax0 * ay0 * az0};
    // accumulation on packed fp32 pairs (v_pk_mul_f32 / v_pk_add_f32: two IEEE-exact operations per instruction, same
    // rounding as the scalar mul and add of the reference -- this TU is built without FMA contraction -- so results stay
    // bit-exact while the 96 mul + 96 add of a fetch become 48 + 48 instructions; the kernel is ~70 % issue-bound)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f acc2[6];
#pragma unroll
    for (int k = 0; k < 6; k++) acc2[k] = (v2f){0.0f, 0.0f};
    // per-axis range flags and ONE 32-bit base index (the grid has < 2^31 voxels, checked at the entry point): the
    // per-corner 64-bit multiply-adds were quarter-rate instructions on an issue-bound kernel
    const bool okx[2] = {x0 >= 0 && x0 < W, x1 >= 0 && x1 < W};
    const bool oky[2] = {y0 >= 0 && y0 < H, y1 >= 0 && y1 < H};
    const bool okz[2] = {z0 >= 0 && z0 < D, z1 >= 0 && z1 < D};
    const int lin0 = vox0 + (z0 * H + y0) * W + x0;
    const int sy = W, sz = H * W;
    // channel-last: the byte offset of corner 0 is computed ONCE and pinned in a register (hipcc otherwise re-derives every corner's
    // voxel index from z0, y0, x0 with two v_mad_u64_u32 and a v_mul_lo_u32 -- 24 quarter-rate instructions per fetch, as much
    // issue time as the 96 packed multiply-adds of the interpolation); the other corners add wave-uniform constants
    uint32_t boff0 = (uint32_t)lin0 * 48u;
    if (LAYOUT == IA_LAYOUT_NDHWC) asm volatile("" : "+v"(boff0));
    if (LAYOUT == IA_LAYOUT_NDHWC && PAIRS) {
        // the two x-corners of a (y, z) edge are 96 contiguous bytes: SIX loads in flight per phase and four phases per fetch instead
        // of three loads and eight phases (every phase ends in a wait for its own loads).  A corner outside the grid reads its
        // in-range neighbour's address with weight 0: acc + (+-0) leaves every accumulator bit unchanged (the accumulators start at
        // +0 and can only become -0 from two -0 addends), so the result is the reference's, which skips such corners.
        const char* base = reinterpret_cast<const char*>(vJ);
        // (a wave-uniform fast path without the selects, for waves whose lanes all have both x-corners in range, measured no gain)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool okyz = oky[e & 1] && okz[(e >> 1) & 1];
            if (okyz && (okx[0] || okx[1])) {
                const uint32_t cst = (uint32_t)((e & 1) * sy + ((e >> 1) & 1) * sz) * 48u;            // scalar
                const uint32_t b0 = boff0 + cst + (okx[0] ? 0u : 48u);                                // corner x0 (or x1's voxel if x0 is outside)
                const uint32_t b1 = boff0 + cst + (okx[1] ? 48u : 0u);                                // corner x1 (or x0's voxel)
                const float w0 = okx[0] ? wgt[2 * e] : 0.0f, w1 = okx[1] ? wgt[2 * e + 1] : 0.0f;
                const float4 a0 = *reinterpret_cast<const float4*>(base + b0);
                const float4 a1 = *reinterpret_cast<const float4*>(base + b0 + 16u);
                const float4 a2 = *reinterpret_cast<const float4*>(base + b0 + 32u);
                const float4 c0 = *reinterpret_cast<const float4*>(base + b1);
                const float4 c1 = *reinterpret_cast<const float4*>(base + b1 + 16u);
                const float4 c2 = *reinterpret_cast<const float4*>(base + b1 + 32u);
                const v2f u0 = (v2f){w0, w0}, u1 = (v2f){w1, w1};
                acc2[0] = acc2[0] + (v2f){a0.x, a0.y} * u0; acc2[1] = acc2[1] + (v2f){a0.z, a0.w} * u0;
                acc2[2] = acc2[2] + (v2f){a1.x, a1.y} * u0; acc2[3] = acc2[3] + (v2f){a1.z, a1.w} * u0;
                acc2[4] = acc2[4] + (v2f){a2.x, a2.y} * u0; acc2[5] = acc2[5] + (v2f){a2.z, a2.w} * u0;
                acc2[0] = acc2[0] + (v2f){c0.x, c0.y} * u1; acc2[1] = acc2[1] + (v2f){c0.z, c0.w} * u1;
                acc2[2] = acc2[2] + (v2f){c1.x, c1.y} * u1; acc2[3] = acc2[3] + (v2f){c1.z, c1.w} * u1;
                acc2[4] = acc2[4] + (v2f){c2.x, c2.y} * u1; acc2[5] = acc2[5] + (v2f){c2.z, c2.w} * u1;
            }
        }
    } else {
#pragma unroll
    for (int c = 0; c < 8; c++) {
        if (okx[c & 1] && oky[(c >> 1) & 1] && okz[(c >> 2) & 1]) {
            float v[12];
            if (LAYOUT == IA_LAYOUT_NDHWC) {
                const uint32_t cst = (uint32_t)((c & 1) + ((c >> 1) & 1) * sy + ((c >> 2) & 1) * sz) * 48u;     // scalar
                const char* base = reinterpret_cast<const char*>(vJ);
                const uint32_t boff = boff0 + cst;
                const float4 a = *reinterpret_cast<const float4*>(base + boff);
                const float4 b = *reinterpret_cast<const float4*>(base + boff + 16u);
                const float4 d = *reinterpret_cast<const float4*>(base + boff + 32u);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                v[8] = d.x; v[9] = d.y; v[10] = d.z; v[11] = d.w;
            } else {
                load_corner<LAYOUT>(vJ, vol, (int64_t)(lin0 + (c & 1) + ((c >> 1) & 1) * sy + ((c >> 2) & 1) * sz), v);
            }
            const v2f w2 = (v2f){wgt[c], wgt[c]};
#pragma unroll
            for (int k = 0; k < 6; k++) acc2[k] = acc2[k] + (v2f){v[2 * k], v[2 * k + 1]} * w2;
        }
    }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) { out[2 * k] = acc2[k].x; out[2 * k + 1] = acc2[k].y; }
}

__device__ __forceinline__ void J_inv_update(float Ji[9], float x0, float x1, float x2, float g0, float g1, float g2)
{
    const float J00 = Ji[0], J01 = Ji[1], J02 = Ji[2], J10 = Ji[3], J11 = Ji[4], J12 = Ji[5], J20 = Ji[6],
                J21 = Ji[7], J22 = Ji[8];
    const float c0 = J00 * x0 + J10 * x1 + J20 * x2;
    const float c1 = J01 * x0 + J11 * x1 + J21 * x2;
    const float c2 = J02 * x0 + J12 * x1 + J22 * x2;
    const float s = c0 * g0 + c1 * g1 + c2 * g2;
    const float r0 = -J00 * g0 - J01 * g1 - J02 * g2;
    const float r1 = -J10 * g0 - J11 * g1 - J12 * g2;
    const float r2 = -J20 * g0 - J21 * g1 - J22 * g2;
    // nine IEEE divisions by the SAME denominator (fuse_cuda_kernel_fast.cu:232-243).  hipcc expands a / s into 11 instructions:
    //   v_div_scale (s) . v_rcp . fma . fma  [reciprocal r1 of s, one Newton step]
    //   v_div_scale (a) . q0 = a r1 . rem0 = fma(-s, q0, a) . q1 = fma(rem0, r1, q0) . rem1 = fma(-s, q1, a) . v_div_fmas . v_div_fixup
    // v_div_scale leaves both operands as they are -- and v_div_fmas is then a plain fma, v_div_fixup returns its first operand --
    // unless an operand or the quotient is zero / denormal / huge, the exponents differ by 96 or more, or the numerator is below
    // 2^-103.  With every |numerator| and |s| in [2^-40, 2^40] none of that can happen, the nine expansions compute the same r1
    // nine times, and sharing it leaves the SAME five operations per quotient: bit-identical, 48 + 15 (range test) instead of 99
    // VALU instructions on a kernel whose VALU is 77 % busy (tools/pmc_probe.sh).  Anything else (NaN included: the comparisons
    // fail) takes the plain divisions.
    const float t0 = r0 + x0, t1 = r1 + x1, t2 = r2 + x2;
    const float n[9] = {c0 * t0, c1 * t0, c2 * t0, c0 * t1, c1 * t1, c2 * t1, c0 * t2, c1 * t2, c2 * t2};
    const float mx = fmaxf(fmaxf(fmaxf(fabsf(n[0]), fabsf(n[1])), fmaxf(fabsf(n[2]), fabsf(n[3]))),
                           fmaxf(fmaxf(fabsf(n[4]), fabsf(n[5])), fmaxf(fmaxf(fabsf(n[6]), fabsf(n[7])), fabsf(n[8]))));
    const float mn = fminf(fminf(fminf(fabsf(n[0]), fabsf(n[1])), fminf(fabsf(n[2]), fabsf(n[3]))),
                           fminf(fminf(fabsf(n[4]), fabsf(n[5])), fminf(fminf(fabsf(n[6]), fabsf(n[7])), fabsf(n[8]))));
    const float as = fabsf(s);
    if (mx <= 0x1p40f && mn >= 0x1p-40f && as <= 0x1p40f && as >= 0x1p-40f) {
        float r = __builtin_amdgcn_rcpf(s);
        r = fmaf(fmaf(-s, r, 1.0f), r, r);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float q = n[k] * r;
            q = fmaf(fmaf(-s, q, n[k]), r, q);
            q = fmaf(fmaf(-s, q, n[k]), r, q);
            Ji[k] += q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) Ji[k] += n[k] / s;
    }
}

// ---- K8 -------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(THREADS) void broyden_kernel(
    int64_t total, int64_t N, int I, const float* __restrict__ xd_tgt, const float* __restrict__ voxel_J, int D, int H,
    int W, const float* __restrict__ tfs, const int32_t* __restrict__ bone_ids, const float* __restrict__ offset_g,
    const float* __restrict__ scale_g, float cvg_threshold, float dvg_threshold, float* __restrict__ x,
    float* __restrict__ J_inv, uint8_t* __restrict__ is_valid, float* __restrict__ fwd_J)
{
    const int64_t index = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (index >= total) return;
    const int64_t vol = (int64_t)D * H * W;
    const int i_batch = (int)(index / (N * I));
    const int64_t i_point = (index % (N * I)) / I;
    const int i_init = (int)((index % (N * I)) % I);
    // channel-last: uniform base + per-item voxel offset; reference layout: per-batch base pointer
    const float* vJ = (LAYOUT == IA_LAYOUT_NDHWC) ? voxel_J : voxel_J + (int64_t)i_batch * 12 * vol;
    const int vox0 = (LAYOUT == IA_LAYOUT_NDHWC) ? (int)(i_batch * vol) : 0;
    const float offset[3] = {offset_g[0], offset_g[1], offset_g[2]};
    const float scale[3] = {scale_g[0], scale_g[1], scale_g[2]};
    float gx[3], gx_new[3] = {0, 0, 0}, xt[3], x_l[3];
    xt[0] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 0];
    xt[1] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 1];
    xt[2] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 2];
    const int i_bone = bone_ids[i_init];
    const float* T = tfs + ((int64_t)i_batch * 24 + i_bone) * 16;
    const float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
    x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
    x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
    x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];

    float Jl[12];
    grid_sample_J<LAYOUT>(vJ, vox0, D, H, W, scale[0] * (x_l[0] + offset[0]), scale[1] * (x_l[1] + offset[1]),
                          scale[2] * (x_l[2] + offset[2]), Jl);
    float Ji[9];
    Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
    Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
    Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];

    for (int it = 0; it < 10; it++) {
        const float J00 = Ji[0], J01 = Ji[1], J02 = Ji[2], J10 = Ji[3], J11 = Ji[4], J12 = Ji[5], J20 = Ji[6],
                    J21 = Ji[7], J22 = Ji[8];
        if (it == 0) {
            gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3];
            gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7];
            gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11];
            gx[0] = gx[0] - xt[0]; gx[1] = gx[1] - xt[1]; gx[2] = gx[2] - xt[2];
        } else {
            gx[0] = gx_new[0]; gx[1] = gx_new[1]; gx[2] = gx_new[2];
        }
        const float u0 = -J00 * gx[0] + -J01 * gx[1] + -J02 * gx[2];
        const float u1 = -J10 * gx[0] + -J11 * gx[1] + -J12 * gx[2];
        const float u2 = -J20 * gx[0] + -J21 * gx[1] + -J22 * gx[2];
        x_l[0] += u0; x_l[1] += u1; x_l[2] += u2;
        const float ix = scale[0] * (x_l[0] + offset[0]);
        const float iy = scale[1] * (x_l[1] + offset[1]);
        const float iz = scale[2] * (x_l[2] + offset[2]);
        grid_sample_J<LAYOUT>(vJ, vox0, D, H, W, ix, iy, iz, Jl);
        gx_new[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
        gx_new[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
        gx_new[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
        const float norm_gx = gx_new[0] * gx_new[0] + gx_new[1] * gx_new[1] + gx_new[2] * gx_new[2];
        if (norm_gx < cvg_threshold * cvg_threshold) {
            const bool ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
            is_valid[index] = ok ? 1 : 0;
            if (ok) {
                x[index * 3 + 0] = x_l[0]; x[index * 3 + 1] = x_l[1]; x[index * 3 + 2] = x_l[2];
                if (J_inv) {
                    float* Jo = J_inv + index * 9;
                    Jo[0] = J00; Jo[1] = J01; Jo[2] = J02; Jo[3] = J10; Jo[4] = J11; Jo[5] = J12;
                    Jo[6] = J20; Jo[7] = J21; Jo[8] = J22;
                }
                if (fwd_J) {   // forward LBS Jacobian at the root == blended bone rotation (fwd_tfs, deformer_torch.py:49-52)
                    float* Fo = fwd_J + index * 9;
                    Fo[0] = Jl[0]; Fo[1] = Jl[1]; Fo[2] = Jl[2]; Fo[3] = Jl[4]; Fo[4] = Jl[5]; Fo[5] = Jl[6];
                    Fo[6] = Jl[8]; Fo[7] = Jl[9]; Fo[8] = Jl[10];
                }
            }
            return;
        } else if (norm_gx > dvg_threshold * dvg_threshold) {
            is_valid[index] = 0;
            return;
        }
        J_inv_update(Ji, u0, u1, u2, gx_new[0] - gx[0], gx_new[1] - gx[1], gx_new[2] - gx[2]);
    }
    is_valid[index] = 0;      // not converged after 10 steps (same value the caller's zero-fill gives; lets internal callers skip the fill)
}

// ---- K8, lane-persistent form ---------------------------------------------------------
// The (point, init) searches have wildly different lengths (most of the 13 bone initialisations diverge after
// 1-3 Broyden steps, the good ones run 4-10), so "one item per lane" leaves a wave idling behind its slowest
// item.  Here each wave owns a contiguous chunk of items and every lane runs a state machine whose body is
// exactly ONE trilinear fetch + its post-processing; a lane that finishes pulls the wave's next item
// (ballot + popcount, wave-private cursor: no atomics, deterministic).  Per-item arithmetic is the same
// operation sequence as broyden_kernel / the oracle, so results stay bit-exact.
constexpr int BR_CHUNK = 1024;      // items per wave

template <int LAYOUT>
__global__ __launch_bounds__(THREADS) void broyden_persistent_kernel(
    int64_t total, int64_t N, int I, const float* __restrict__ xd_tgt, const float* __restrict__ voxel_J, int D, int H,
    int W, const float* __restrict__ tfs, const int32_t* __restrict__ bone_ids, const float* __restrict__ offset_g,
    const float* __restrict__ scale_g, float cvg_threshold, float dvg_threshold, float* __restrict__ x,
    float* __restrict__ J_inv, uint8_t* __restrict__ is_valid, float* __restrict__ fwd_J, int br_chunk)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = ((int64_t)blockIdx.x * THREADS + threadIdx.x) >> 6;
    int64_t cursor = wave_id * br_chunk;                                  // wave-uniform
    const int64_t chunk_end = (cursor + br_chunk < total) ? cursor + br_chunk : total;
    if (cursor >= total) return;
    const int64_t vol = (int64_t)D * H * W;
    const float offset[3] = {offset_g[0], offset_g[1], offset_g[2]};
    const float scale[3] = {scale_g[0], scale_g[1], scale_g[2]};
    const float cvg2 = cvg_threshold * cvg_threshold, dvg2 = dvg_threshold * dvg_threshold;

    bool have = false;
    int64_t index = 0;
    int it = -1;                  // -1: waiting for the initial fetch
    float xt[3] = {0, 0, 0}, x_l[3] = {0, 0, 0}, gx[3] = {0, 0, 0}, u[3] = {0, 0, 0};
    float Ji[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* vJ = voxel_J;
    int vox0 = 0;

    for (;;) {
        // ---- refill idle lanes from the wave's chunk ----
        const unsigned long long need = __ballot(!have);
        if (need) {
            const int rank = __popcll(need & ((1ull << lane) - 1ull));
            if (!have) {
                const int64_t cand = cursor + rank;
                if (cand < chunk_end) {
                    have = true;
                    index = cand;
                    it = -1;
                    const int i_batch = (int)(index / (N * I));
                    const int64_t i_point = (index % (N * I)) / I;
                    const int i_init = (int)((index % (N * I)) % I);
                    if (LAYOUT == IA_LAYOUT_NDHWC) vox0 = (int)(i_batch * vol); else vJ = voxel_J + (int64_t)i_batch * 12 * vol;
                    xt[0] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 0];
                    xt[1] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 1];
                    xt[2] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 2];
                    const float* T = tfs + ((int64_t)i_batch * 24 + bone_ids[i_init]) * 16;
                    const float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
                    x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
                    x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
                    x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];
                }
            }
            cursor += __popcll(need);
            if (cursor > chunk_end) cursor = chunk_end;
        }
        if (!__any(have)) break;
        if (!have) continue;
        // ---- one fetch at the current x_l ----
        const float ix = scale[0] * (x_l[0] + offset[0]);
        const float iy = scale[1] * (x_l[1] + offset[1]);
        const float iz = scale[2] * (x_l[2] + offset[2]);
        float Jl[12];
        grid_sample_J<LAYOUT>(vJ, vox0, D, H, W, ix, iy, iz, Jl);
        if (it < 0) {
            // initial fetch: J_inv guess and g(x0)   (fuse_cuda_kernel_fast.cu:295-331)
            Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
            Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
            Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];
            gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3];
            gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7];
            gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11];
            gx[0] = gx[0] - xt[0]; gx[1] = gx[1] - xt[1]; gx[2] = gx[2] - xt[2];
            it = 0;
        } else {
            // fetch of iteration `it` (x_l already updated): residual, tests, Broyden update (:352-411)
            float gn[3];
            gn[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
            gn[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
            gn[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
            const float norm_gx = gn[0] * gn[0] + gn[1] * gn[1] + gn[2] * gn[2];
            if (norm_gx < cvg2) {
                const bool ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
                is_valid[index] = ok ? 1 : 0;
                if (ok) {
                    x[index * 3 + 0] = x_l[0]; x[index * 3 + 1] = x_l[1]; x[index * 3 + 2] = x_l[2];
                    if (J_inv) {
                        float* Jo = J_inv + index * 9;
#pragma unroll
                        for (int k = 0; k < 9; k++) Jo[k] = Ji[k];
                    }
                    if (fwd_J) {
                        float* Fo = fwd_J + index * 9;
                        Fo[0] = Jl[0]; Fo[1] = Jl[1]; Fo[2] = Jl[2]; Fo[3] = Jl[4]; Fo[4] = Jl[5]; Fo[5] = Jl[6];
                        Fo[6] = Jl[8]; Fo[7] = Jl[9]; Fo[8] = Jl[10];
                    }
                }
                have = false;
                continue;
            } else if (norm_gx > dvg2) {
                is_valid[index] = 0;
                have = false;
                continue;
            }
            J_inv_update(Ji, u[0], u[1], u[2], gn[0] - gx[0], gn[1] - gx[1], gn[2] - gx[2]);
            gx[0] = gn[0]; gx[1] = gn[1]; gx[2] = gn[2];
            it++;
            if (it >= 10) { is_valid[index] = 0; have = false; continue; }       // not converged
        }
        // step: update = -J_inv g, x += update
        u[0] = -Ji[0] * gx[0] + -Ji[1] * gx[1] + -Ji[2] * gx[2];
        u[1] = -Ji[3] * gx[0] + -Ji[4] * gx[1] + -Ji[5] * gx[2];
        u[2] = -Ji[6] * gx[0] + -Ji[7] * gx[1] + -Ji[8] * gx[2];
        x_l[0] += u[0]; x_l[1] += u[1]; x_l[2] += u[2];
    }
}



// ---- K8, lane-persistent form, v2 (B == 1, channel-last grid) -----------------------------------------------------------
// Same state machine and per-item arithmetic as broyden_persistent_kernel.  What changed is the cost of everything AROUND a
// fetch: ~15 of 64 lanes finish per iteration, so the refill path runs (for the whole wave) on practically every
// iteration.  It used to be two 64-bit divisions + fifteen global loads (target point, the bone's 4x4) per refill; now an
// item is (chunk-local 32-bit counter) -> (point, init) by one multiply-high with a host-computed magic, and the bones'
// rows (the 12 floats x0 = R^T (xd - t) needs) sit in LDS.  Results are bit-identical (the golden test runs all schedules).
#ifndef IA_BR2_WAVES
#define IA_BR2_WAVES 5
#endif
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(IA_BR2_WAVES, IA_BR2_WAVES))) void broyden_persistent2_kernel(
    int64_t total, int I, uint32_t magic_I, const float* __restrict__ xd_tgt, const float* __restrict__ voxel_J, int D, int H, int W,
    const float* __restrict__ tfs, const int32_t* __restrict__ bone_ids, const float* __restrict__ offset_g,
    const float* __restrict__ scale_g, float cvg_threshold, float dvg_threshold, float* __restrict__ x,
    float* __restrict__ J_inv, uint8_t* __restrict__ is_valid, float* __restrict__ fwd_J, int br_chunk)
{
    __shared__ float s_T[16 * 12];                     // per init: rows 0..2 of its bone's 4x4 (R | t)
    for (int t = threadIdx.x; t < I * 12; t += THREADS) s_T[t] = tfs[(int64_t)bone_ids[t / 12] * 16 + (t % 12)];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = ((int64_t)blockIdx.x * THREADS + threadIdx.x) >> 6;
    const int64_t chunk_begin = wave_id * br_chunk;
    if (chunk_begin >= total) return;
    const int n_items = (int)((chunk_begin + br_chunk < total) ? br_chunk : total - chunk_begin);
    const int64_t p0 = chunk_begin / I;                // wave-uniform 64-bit division, once
    const int i0 = (int)(chunk_begin - p0 * I);
    int cur = 0;                                       // items of the chunk handed out so far (wave-uniform)
    const float offset[3] = {offset_g[0], offset_g[1], offset_g[2]};
    const float scale[3] = {scale_g[0], scale_g[1], scale_g[2]};
    const float cvg2 = cvg_threshold * cvg_threshold, dvg2 = dvg_threshold * dvg_threshold;

    bool have = false;
    int64_t index = 0;
    int it = -1;                  // -1: waiting for the initial fetch
    float xt[3] = {0, 0, 0}, x_l[3] = {0, 0, 0}, gx[3] = {0, 0, 0}, u[3] = {0, 0, 0};
    float Ji[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

    for (;;) {
        // ---- refill idle lanes from the wave's chunk ----
        const unsigned long long need = __ballot(!have);
        if (need) {
            if (!have) {
                const int rank = __popcll(need & ((1ull << lane) - 1ull));
                const int c = cur + rank;
                if (c < n_items) {
                    have = true;
                    it = -1;
                    const uint32_t t = (uint32_t)(i0 + c);               // < br_chunk + I <= 4096 + 16
                    const uint32_t q = __umulhi(t, magic_I);             // t / I (magic = ceil(2^32 / I), exact for t < 2^28)
                    const int i_init = (int)(t - q * (uint32_t)I);
                    const int64_t i_point = p0 + q;
                    index = chunk_begin + c;
                    xt[0] = xd_tgt[i_point * 3 + 0];
                    xt[1] = xd_tgt[i_point * 3 + 1];
                    xt[2] = xd_tgt[i_point * 3 + 2];
                    const float* T = s_T + i_init * 12;                  // T[r*4 + c], r < 3
                    const float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
                    x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
                    x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
                    x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];
                }
            }
            cur += __popcll(need);
            if (cur > n_items) cur = n_items;
        }
        if (!__any(have)) break;
        if (!have) continue;
        // ---- one fetch at the current x_l ----
        const float ix = scale[0] * (x_l[0] + offset[0]);
        const float iy = scale[1] * (x_l[1] + offset[1]);
        const float iz = scale[2] * (x_l[2] + offset[2]);
        float Jl[12];
        grid_sample_J<IA_LAYOUT_NDHWC>(voxel_J, 0, D, H, W, ix, iy, iz, Jl);
        if (it < 0) {
            // initial fetch: J_inv guess and g(x0)   (fuse_cuda_kernel_fast.cu:295-331)
            Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
            Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
            Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];
            gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3];
            gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7];
            gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11];
            gx[0] = gx[0] - xt[0]; gx[1] = gx[1] - xt[1]; gx[2] = gx[2] - xt[2];
            it = 0;
        } else {
            // fetch of iteration `it` (x_l already updated): residual, tests, Broyden update (:352-411)
            float gn[3];
            gn[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
            gn[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
            gn[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
            const float norm_gx = gn[0] * gn[0] + gn[1] * gn[1] + gn[2] * gn[2];
            if (norm_gx < cvg2) {
                const bool ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
                is_valid[index] = ok ? 1 : 0;
                if (ok) {
                    x[index * 3 + 0] = x_l[0]; x[index * 3 + 1] = x_l[1]; x[index * 3 + 2] = x_l[2];
                    if (J_inv) {
                        float* Jo = J_inv + index * 9;
#pragma unroll
                        for (int k = 0; k < 9; k++) Jo[k] = Ji[k];
                    }
                    if (fwd_J) {
                        float* Fo = fwd_J + index * 9;
                        Fo[0] = Jl[0]; Fo[1] = Jl[1]; Fo[2] = Jl[2]; Fo[3] = Jl[4]; Fo[4] = Jl[5]; Fo[5] = Jl[6];
                        Fo[6] = Jl[8]; Fo[7] = Jl[9]; Fo[8] = Jl[10];
                    }
                }
                have = false;
                continue;
            } else if (norm_gx > dvg2) {
                is_valid[index] = 0;
                have = false;
                continue;
            }
            J_inv_update(Ji, u[0], u[1], u[2], gn[0] - gx[0], gn[1] - gx[1], gn[2] - gx[2]);
            gx[0] = gn[0]; gx[1] = gn[1]; gx[2] = gn[2];
            it++;
            if (it >= 10) { is_valid[index] = 0; have = false; continue; }       // not converged
        }
        // step: update = -J_inv g, x += update
        u[0] = -Ji[0] * gx[0] + -Ji[1] * gx[1] + -Ji[2] * gx[2];
        u[1] = -Ji[3] * gx[0] + -Ji[4] * gx[1] + -Ji[5] * gx[2];
        u[2] = -Ji[6] * gx[0] + -Ji[7] * gx[1] + -Ji[8] * gx[2];
        x_l[0] += u[0]; x_l[1] += u[1]; x_l[2] += u[2];
    }
}

// ---- K8 + K9-consistent early filter (product path; B == 1, channel-last grid) -----------------------------------------------
// 13 searches per point yield ~1.2 surviving roots: ~58 % of the searches converge, most of them onto a root that a LATER
// init also finds -- and K9 (filter.cu:10-54) drops every root that has a later one within 1e-4.  Here one lane owns one POINT and
// walks its inits in REVERSE order (I-1 .. 0), so when init i runs every later init has already finished and the roots they
// converged to are known exactly.  A search is RETIRED (its remaining fetches never issued) before a fetch at x_k when all of:
//   (a) |x_k - r|_inf < eps for a recorded root r of a later init;
//   (b) r is TIGHT: the Frobenius norm of Broyden's J_inv at r is <= SPEC_TAU.  Every search that converges to the true root
//       behind r stops inside {|g| < cvg} around it, i.e. within ~cvg |J^-1| of it: for a tight root all of them end within a few
//       1e-5 of r -- inside K9's radius, so K9 would drop the retired search whatever its exact end point;
//   (c) x_k lies in the SAME voxel cell as r (shrunk by SPEC_CELL_MARGIN): g is piecewise polynomial with kinks on the cell faces, and
//       two distinct well-conditioned roots closer than eps only occur across a kink -- UNLESS (c') the true Jacobian is tight with one
//       sign of det on r's cell and its 26 neighbours (bit 1 of cell_tight): a coherently oriented piecewise-smooth map is locally
//       injective across the faces too, and the box is the plain eps-box.  The cut costs 10 % of the kernel's time through the schedule
//       (duplicates near a face run to their end: slow points a chunk waits for); with (c') 80 % of the cells do without it:
//       search + rows 8.86 -> 8.08 ms per 16.4 M points, same box, candidate sets unchanged;
//   (d) the search's own J_inv estimate has norm <= SPEC_TAU_SELF (k >= 1): it is not sliding along a near-singular valley, where
//       it could stop farther than 1e-4 from r;
//   (e) the TRUE Jacobian of the skinning map is tight all over r's cell (cell_tight, cell_tightness_kernel above): Broyden's estimate
//       of (b) is blind to a fold of the map next to r, where two roots sit 1e-4 ... 1e-3 apart in a flat valley of |g| < cvg.
// (a)-(c), (e) are ONE box test per recorded root: the eps-box around r intersected with r's cell, empty for a root that is not tight.
// A search that COMPLETES valid is compared (L2, K9's expression) with the recorded roots: below 1e-4 it is a duplicate K9 drops;
// from 2e-4 up it is recorded (K9 keeps it: every later valid root, retired ones included, lies within 1e-4 of a recorded one);
// in between -- or when the SPEC_ROOTS slots are full -- the lane REDOES THE POINT with the filter off (all 13 searches to their
// end) and hands the 13 results to rows_flagged_kernel, which applies K9 literally.  Measured on the march points of the eight reference
// poses (tools/spec_search_probe.py, profiles/r04_spec_search_probe_poses.jsonl; 145.8 M points): 29 ... 41 % fewer fetches than the
// exact search, candidate set identical to K9's on EVERY point (without (e): 1 ... 12 points per pose differ, 3e-7), 1.8e-4 of the
// points redone.  eps: larger boxes are not safe (2e-3 / 3e-3 / 5e-3: 1 / 3 / 6 differing points on one of the poses) and buy little
// (-11 % fetches, -3 % time at 5e-3).  Everything that is not retired runs the operation sequence of broyden_kernel: surviving
// candidates are bit-identical to the exact search's.  eps = 0 never retires.  Scheduling as broyden_persistent2_kernel: every
// loop iteration is exactly one trilinear fetch per busy lane; a lane that finishes a search starts its point's next init at
// once, a lane that finishes a point pulls the workgroup's next point.
__device__ __forceinline__ unsigned in_range_corner_count(float gx, float gy, float gz, int D, int H, int W)
{
    float ix = ((gx + 1.f) / 2) * (W - 1), iy = ((gy + 1.f) / 2) * (H - 1), iz = ((gz + 1.f) / 2) * (D - 1);
    if (ix > 2147483646.0f || ix < -2147483648.0f || !isfinite(ix)) ix = -100.0f;
    if (iy > 2147483646.0f || iy < -2147483648.0f || !isfinite(iy)) iy = -100.0f;
    if (iz > 2147483646.0f || iz < -2147483648.0f || !isfinite(iz)) iz = -100.0f;
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
    const unsigned cx = (unsigned)(x0 >= 0 && x0 < W) + (unsigned)(x0 + 1 >= 0 && x0 + 1 < W);
    const unsigned cy = (unsigned)(y0 >= 0 && y0 < H) + (unsigned)(y0 + 1 >= 0 && y0 + 1 < H);
    const unsigned cz = (unsigned)(z0 >= 0 && z0 < D) + (unsigned)(z0 + 1 >= 0 && z0 + 1 < D);
    return cx * cy * cz;
}

#ifdef IA_SPEC_DIAG_CELLS
#define IA_DIAG_CELLS_ON 1
#else
#define IA_DIAG_CELLS_ON 0
#endif
constexpr int SPEC_ROOTS = 3;              // recorded roots per point (survivors per point average 1.2; a 4th sends the point to the exact redo)
#ifndef IA_SPEC_TAU                        // (-D overrides: rule sweeps through tools/ab_build.sh)
#define IA_SPEC_TAU 2.5f
#endif
#ifndef IA_SPEC_TAU_SELF
#define IA_SPEC_TAU_SELF 3.0f
#endif
constexpr float SPEC_TAU = IA_SPEC_TAU;           // a root is tight when |J_inv|_F <= SPEC_TAU (a rotation has sqrt(3) = 1.73)
constexpr float SPEC_TAU_SELF = IA_SPEC_TAU_SELF; // a search may be retired while its own |J_inv|_F <= SPEC_TAU_SELF
constexpr float SPEC_CELL_MARGIN = 4e-6f;  // the cell box of a root is shrunk by this much (canonical metres) on every side
// points per launch that can be redone exactly (1.4e-4 .. 1.8e-4 of the points are on the march distributions; the list holds 1 / 64 of a batch)
// (at least 2^18 records, 52 MB: a SMALL batch -- broyden_items_rows_kernel below -- sends every point through the list)
static inline int64_t spec_flag_cap(int64_t N) { const int64_t c = N >> 6; return c < (1 << 18) ? (1 << 18) : c; }

// flagged points: the 13 results of the exact redo, for rows_flagged_kernel
struct SpecFlag {
    int32_t* count;      // [1]
    int32_t* point;      // [cap]
    uint32_t* valid;     // [cap] bit i: init i converged inside the box
    float* x;            // [cap][16][3]
    int cap;
};

// PACK: the candidate bookkeeping of the caller done here: the k-th recorded root of a point (k = 0 is its highest init) goes to
// x [N, SPEC_ROOTS, 3] slot k, and the lane leaves cnt[point] and meta[point] = the inits of slots 0..2 in bytes 0..2: 44 bytes per
// point instead of 169 (x [N,I,3] + is_valid [N,I]), no filter pass.  Flagged points leave cnt = 0 and their record; rows_flagged_kernel
// fills in their rows (and overflow records for a 4th, 5th ... survivor).
template <bool COUNT, bool PACK, int WG = THREADS>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(IA_BR2_WAVES, IA_BR2_WAVES))) void broyden_spec_kernel(
    int64_t N, int I, const float* __restrict__ xd_tgt, const float* __restrict__ voxel_J, int D, int H, int W,
    const float* __restrict__ tfs, const int32_t* __restrict__ bone_ids, const float* __restrict__ offset_g,
    const float* __restrict__ scale_g, float cvg_threshold, float dvg_threshold, float eps, float* __restrict__ x,
    float* __restrict__ J_inv, uint8_t* __restrict__ is_valid, float* __restrict__ fwd_J, int pts_per_wave,
    unsigned long long* __restrict__ counters /* NULL or [5]: fetches, retired items, completed valid items, points redone exactly, in-range corner loads */,
    int32_t* __restrict__ cnt /* PACK: [N] */, uint32_t* __restrict__ meta /* PACK: [N] */, SpecFlag flag /* PACK */,
    int slots /* <= SPEC_ROOTS: roots recorded / row slots used (test hook) */,
    const int32_t* __restrict__ order /* NULL or [N]: point p of this launch is xd_tgt[order[p]] (evaluation order != storage order) */,
    const uint8_t* __restrict__ cell_tight /* NULL or [D,H,W] (ia_cell_tightness): a root in a cell with 0 gets no retirement box */)
{
    __shared__ float s_T[16 * 12];                     // per init: rows 0..2 of its bone's 4x4 (R | t)
    __shared__ int s_cur;                              // positions of the WORKGROUP's stream of chunks handed out so far
    for (int t = threadIdx.x; t < I * 12; t += WG) s_T[t] = tfs[(int64_t)bone_ids[t / 12] * 16 + (t % 12)];
    if (threadIdx.x == 0) s_cur = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // the four waves of a workgroup pull points off ONE chunk (pts_per_wave x 4 consecutive points of the sorted order): the
    // lanes that are busy at any time sit in a compact window of it (what the L1 sees), and the drain at the chunk's end -- lanes
    // idle because nothing is left to pull -- is paid once per 4 x pts_per_wave points.  Headline step (male-3-casual:0), ms:
    // private chunks of 80 / 160 / 320 / 640 / 1280 / 4096 points per wave: 412 / 377 / 368 / 384 / 409 / 493 (long chunks: the resident
    // waves of the device work too far apart in the sorted order for the L2s); shared chunks of 4 x 160 / 320 / 640 / 1280: 367 / 369 / 384 / 405.
    // MORE waves on one chunk do not help (same step, 192 points per wave, after the head's speed-up): workgroups of 4 / 8 / 10 / 16
    // waves 355.8 / 367.4 / 400.7 / 377.5 ms; 16 waves x 96 points 371.3 -- the CU's L1 is not what the window buys, the L2s are.
    // Occupancy: 4 instead of 5 waves per SIMD (94 VGPRs either way) 359.8 against 352.7 ms; 6 waves (80 VGPRs: 26 spilled dwords in the loop) 480.4;
    // 6 waves with the recorded roots, the target point and the PACK counters in LDS (86 VGPRs unconstrained, 10 dwords still spilled at
    // 80): search 187 against 143 ms -- a handful of scratch reloads per fetch cost more than the sixth wave hides
    // Workgroup b works through the chunks b, b + G, b + 2 G ... (G = grid size) as ONE stream: the LDS cursor runs on across chunk
    // boundaries.  Position q of the stream is point ((q >> log2c) G + b) 2^log2c + (q & mask): increasing in q, so the first position
    // at or beyond N ends the stream.  The product launches G = number of chunks (one chunk per workgroup, one point per lane: see
    // launch_spec for the measurements; a grid of resident workgroups -- persistent, no drain -- is slower).
    const int log2c = pts_per_wave;                    // log2 of the points per chunk (per workgroup)
    const int cmask = (1 << log2c) - 1;
    const int G = gridDim.x, b_wg = blockIdx.x;
    if (((int64_t)b_wg << log2c) >= N) return;
    auto stream_point = [&](int q) -> int64_t { return (((int64_t)(q >> log2c) * G + b_wg) << log2c) + (q & cmask); };
    bool drained = false;                              // wave-uniform: the chunk has nothing left
    const float offset[3] = {offset_g[0], offset_g[1], offset_g[2]};
    const float scale[3] = {scale_g[0], scale_g[1], scale_g[2]};
    const float cvg2 = cvg_threshold * cvg_threshold, dvg2 = dvg_threshold * dvg_threshold;
    // voxel cell f of axis a covers canonical x in [f cell_w + cell_o, (f + 1) cell_w + cell_o)   (g = scale (x + offset), f = floor(((g + 1) / 2) (dim - 1))).
    // The twelve wave-uniform constants of the cell bounds live in LDS (the kernel has no scalar register left for them: as SGPR
    // candidates they ended up in vector registers and were spilled to scratch -- reloaded in a block that runs on almost every iteration):
    // s_cell[0..2] pitch, [3..5] / [6..8] lower / upper bound of cell 0 shrunk by the margin (a negative pitch swaps the ends), [9..11] dim - 1
    __shared__ float s_cell[12];
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const int dim_a = a == 0 ? W : (a == 1 ? H : D);
        const float cw = 2.0f / ((dim_a - 1) * scale_g[a]);
        const float co = -1.0f / scale_g[a] - offset_g[a];
        s_cell[a] = cw;
        s_cell[3 + a] = co + fminf(cw, 0.0f) + SPEC_CELL_MARGIN;
        s_cell[6 + a] = co + fmaxf(cw, 0.0f) - SPEC_CELL_MARGIN;
        s_cell[9 + a] = (float)(dim_a - 1);
    }
    __syncthreads();

    bool have = false;            // lane owns a point
    bool next = false;            // current search ended: move to the point's next init (or give the point up)
    int pt = 0;                   // the lane's point (index into the launch's N points)
    int init = 0;
    int it = -1;                  // -1: waiting for the initial fetch
    int n_roots = 0;              // recorded roots of the lane's point (stays 0 while the point is redone exactly)
    // rarely touched per-lane state lives in LDS: [0] roots recorded as candidates (PACK) or -1 = the point is being REDONE EXACTLY,
    // [1] their inits, one byte each (exact redo: the valid mask of the 13 searches), [2] the point's record in the flagged list
    __shared__ int s_state[3][WG];
#define n_done s_state[0][threadIdx.x]
#define inits (reinterpret_cast<unsigned*>(s_state[1])[threadIdx.x])
#define flag_rec s_state[2][threadIdx.x]
    n_done = 0; inits = 0; flag_rec = -1;
    float xt[3] = {0, 0, 0}, x_l[3] = {0, 0, 0}, gx[3] = {0, 0, 0}, u[3] = {0, 0, 0};
    float Ji[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // per recorded root: the root (for K9's distance when a later search completes) and its RETIREMENT BOX = eps-box around it
    // intersected with its (shrunk) voxel cell; lo = +inf for a root that is not tight
    __shared__ float s_root[SPEC_ROOTS * 9][WG];
    float* const rootp = &s_root[0][threadIdx.x];
#define ROOT(r, k) rootp[((r) * 9 + (k)) * WG]
#define BOXLO(r, k) rootp[((r) * 9 + 3 + (k)) * WG]
#define BOXHI(r, k) rootp[((r) * 9 + 6 + (k)) * WG]
    unsigned c_fetch = 0, c_retired = 0, c_valid = 0, c_redo = 0, c_corner = 0;

    for (;;) {
        // ---- transitions: next init of the lane's point, or the wave's next point ----
        if (have && next) {
            next = false;
            if (init > 0) { init--; it = -1; }
            else {
                have = false;
                if (PACK) {
                    if (n_done < 0) {                                    // exact redo finished: hand the 13 results over
                        cnt[pt] = 0;
                        meta[pt] = 0u;
                        if (flag_rec >= 0) { flag.point[flag_rec] = pt; flag.valid[flag_rec] = inits; }
                    } else {
                        cnt[pt] = n_done;
                        meta[pt] = inits;
                    }
                }
            }
        }
        // (measured: letting 4 .. 32 lanes go idle before the wave runs the refill below -- ~90 VALU instructions + dependent loads for
        //  one or two lanes on most iterations -- is slower, 147 .. 149 against 144.7 ms per headline step: idle lanes cost more)
        const unsigned long long need = drained ? 0ull : __ballot(!have);
        if (need) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_cur, __popcll(need));
            base = __builtin_amdgcn_readfirstlane(base);
            drained = stream_point(base + __popcll(need)) >= N;
            if (!have) {
                const int rank = __popcll(need & ((1ull << lane) - 1ull));
                const int64_t p = stream_point(base + rank);
                if (p < N) {
                    have = true;
                    pt = (int)p;
                    init = I - 1;
                    it = -1;
                    n_roots = 0;
                    n_done = 0;
                    inits = 0;
                    flag_rec = -1;
                    const int64_t src = order ? (int64_t)order[p] : p;
                    xt[0] = xd_tgt[src * 3 + 0];
                    xt[1] = xd_tgt[src * 3 + 1];
                    xt[2] = xd_tgt[src * 3 + 2];
                }
            }
        }
        if (!__any(have)) { if (drained) break; else continue; }
        if (!have) continue;
        if (it < 0) {                                                    // a new search: x0 = R^T (xd - t) of its bone
            const float* T = s_T + init * 12;                            // T[r*4 + c], r < 3
            const float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
            x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
            x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
            x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];
        }
        const int64_t index = (int64_t)pt * I + init;
        if (it < 0) {
            // the rigid inverse of a bone that owns the point IS the root (up to rounding): a search that STARTS inside the retirement
            // box of a root a later init has found is retired before its first fetch
            bool at_root = false;
#pragma unroll
            for (int r = 0; r < SPEC_ROOTS; r++) {
                if (r < n_roots) {
                    // all six bounds loaded up front, the comparisons combined bitwise: written with && the compiler short-circuits -- six
                    // nested branches, each waiting for its own LDS read
                    const float l0 = BOXLO(r, 0), l1 = BOXLO(r, 1), l2 = BOXLO(r, 2), h0 = BOXHI(r, 0), h1 = BOXHI(r, 1), h2 = BOXHI(r, 2);
                    at_root = at_root | ((x_l[0] >= l0) & (x_l[0] < h0) & (x_l[1] >= l1) & (x_l[1] < h1) & (x_l[2] >= l2) & (x_l[2] < h2));
                }
            }
            if (at_root) { if (!PACK) is_valid[index] = 0; if (COUNT && !IA_DIAG_CELLS_ON) c_retired++; next = true; }
        }
        // a lane retired here sits this fetch out (it starts its next search in the next iteration)
        if (next) continue;
        // ---- one fetch at the current x_l ----
        const float ix = scale[0] * (x_l[0] + offset[0]);
        const float iy = scale[1] * (x_l[1] + offset[1]);
        const float iz = scale[2] * (x_l[2] + offset[2]);
        float Jl[12];
        grid_sample_J<IA_LAYOUT_NDHWC, true>(voxel_J, 0, D, H, W, ix, iy, iz, Jl);
#ifdef IA_SPEC_DIAG_ITERS       /* diagnostic build: counters[4] = lane-slots of the fetch iterations (64 per wave iteration that fetched) */
        if (COUNT) { c_fetch++; c_corner += (__ffsll((long long)__ballot(1)) - 1 == lane) ? 64u : 0u; }      // exact: one lane of the wave adds the 64 slots
#ifdef IA_SPEC_DIAG_CELLS       /* counters[1] = fetches in the voxel cell of the wave's first active lane, counters[2] = ... of the first lane outside that cell */
        if (COUNT) {
            const float fx_ = ((ix + 1.f) / 2) * (W - 1), fy_ = ((iy + 1.f) / 2) * (H - 1), fz_ = ((iz + 1.f) / 2) * (D - 1);
            const bool fin = fabsf(fx_) < 1e6f && fabsf(fy_) < 1e6f && fabsf(fz_) < 1e6f;
            const int cell = fin ? (((int)floorf(fz_) + 512) << 20) | (((int)floorf(fy_) + 512) << 10) | ((int)floorf(fx_) + 512) : -1 - lane;
            const int lead = __builtin_amdgcn_readfirstlane(cell);
            const unsigned long long rest = __ballot(cell != lead);
            c_retired += (cell == lead) ? 1u : 0u;
            if (rest) {
                const int second = __shfl(cell, __ffsll((long long)rest) - 1, 64);
                c_valid += (cell == second) ? 1u : 0u;
            }
        }
#endif
#else
        if (COUNT) { c_fetch++; c_corner += in_range_corner_count(ix, iy, iz, D, H, W); }
#endif
        if (it < 0) {
            // initial fetch: J_inv guess and g(x0)   (fuse_cuda_kernel_fast.cu:295-331)
            Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
            Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
            Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];
            gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3];
            gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7];
            gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11];
            gx[0] = gx[0] - xt[0]; gx[1] = gx[1] - xt[1]; gx[2] = gx[2] - xt[2];
            it = 0;
        } else {
            // fetch of iteration `it` (x_l already updated): residual, tests, Broyden update (:352-411)
            float gn[3];
            gn[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
            gn[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
            gn[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
            const float norm_gx = gn[0] * gn[0] + gn[1] * gn[1] + gn[2] * gn[2];
            if (norm_gx < cvg2) {
                const bool ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
                if (!PACK) is_valid[index] = ok ? 1 : 0;
                if (ok) {
                    if (COUNT && !IA_DIAG_CELLS_ON) c_valid++;
                    if (!PACK) { x[index * 3 + 0] = x_l[0]; x[index * 3 + 1] = x_l[1]; x[index * 3 + 2] = x_l[2]; }
                    if (J_inv) {
                        float* Jo = J_inv + index * 9;
#pragma unroll
                        for (int k = 0; k < 9; k++) Jo[k] = Ji[k];
                    }
                    if (fwd_J) {
                        float* Fo = fwd_J + index * 9;
                        Fo[0] = Jl[0]; Fo[1] = Jl[1]; Fo[2] = Jl[2]; Fo[3] = Jl[4]; Fo[4] = Jl[5]; Fo[5] = Jl[6];
                        Fo[6] = Jl[8]; Fo[7] = Jl[9]; Fo[8] = Jl[10];
                    }
                    if (n_done < 0) {
                        // exact redo of a flagged point: every result goes to the point's record (PACK), nothing is recorded as a root
                        if (PACK) {
                            inits |= 1u << init;
                            if (flag_rec >= 0) {
                                float* fx = flag.x + ((int64_t)flag_rec * 16 + init) * 3;
                                fx[0] = x_l[0]; fx[1] = x_l[1]; fx[2] = x_l[2];
                            }
                        }
                    } else {
                        // K9 against the recorded roots (filter.cu:38-45: squared L2 distance below 1e-4^2 drops the earlier init)
                        float dmin = 1e30f;
#pragma unroll
                        for (int r = 0; r < SPEC_ROOTS; r++) {
                            if (r < n_roots) {
                                const float d0 = x_l[0] - ROOT(r, 0), d1 = x_l[1] - ROOT(r, 1), d2 = x_l[2] - ROOT(r, 2);
                                dmin = fminf(dmin, d0 * d0 + d1 * d1 + d2 * d2);
                            }
                        }
                        // (double)dist < 0.0001 * 0.0001 for a float dist <=> dist < the smallest float above that double (0x322bcc78)
                        const bool dup = dmin < __uint_as_float(0x322bcc78u);
                        if (!dup && (dmin < 4e-8f || n_roots >= slots)) {
                            // neither surely a duplicate nor surely distinct from every later valid root (or no slot left): the lane
                            // searches this point again with the filter off and lets rows_flagged_kernel apply K9 to all 13 results
                            if (COUNT) c_redo++;
                            n_done = -1;
                            inits = 0;
                            n_roots = 0;
                            if (PACK) {
                                const int k = atomicAdd(flag.count, 1);
                                flag_rec = k < flag.cap ? k : -1;            // a full list is reported through the count
                            }
                            init = I - 1;
                            it = -1;
                            continue;
                        }
                        if (!dup) {
                            if (PACK) {
                                float* const row = x + ((int64_t)pt * SPEC_ROOTS + n_done) * 3;
                                row[0] = x_l[0]; row[1] = x_l[1]; row[2] = x_l[2];
                                inits |= (unsigned)init << (8 * n_done);
                                n_done++;
                            }
                            // record the root with its retirement box: the eps-box cut to the root's voxel cell; empty unless tight.
                            // What each condition costs (same-box A/B, tools/ab_build.sh + tools/search_ab.py, 16.4 M headline march points,
                            // search + rows ms; round 3's unconditional eps test: 8.34): eps box only (-DIA_SPEC_ABL_EPS_BOX) 8.54; + tightness
                            // (-DIA_SPEC_ABL_X3) 8.55; cut COMPUTED but unused (-DIA_SPEC_ABL_X1) 8.49; cut used (this build) 9.41 -- the arithmetic
                            // is free, the cut itself costs 10 %: with it 90.6 instead of 95.1 % of the lanes are busy at a fetch
                            // (-DIA_SPEC_DIAG_ITERS) for 1 % more fetches.  It is what takes the candidate-set differences from 5 to 1 of 16.4 M points.
                            // This block is entered by SOME lane of the wave on almost every iteration (~2.6 roots are recorded per wave and
                            // iteration), so every instruction in it costs like one of the main loop's: ~90 VALU as first written took the
                            // search from 8.9 to 10.9 ms per 16.4 M points (same-box A/B, tools/search_ab.py).  Cell bounds are two fma off
                            // wave-uniform constants (margin and the sign of the cell pitch folded in), min / max as selects, tightness
                            // applied to one axis (a box with one empty side is empty).
                            const float jn2 = fmaf(Ji[8], Ji[8], fmaf(Ji[7], Ji[7], fmaf(Ji[6], Ji[6], fmaf(Ji[5], Ji[5], fmaf(Ji[4], Ji[4],
                                              fmaf(Ji[3], Ji[3], fmaf(Ji[2], Ji[2], fmaf(Ji[1], Ji[1], Ji[0] * Ji[0]))))))));
                            const float gc[3] = {ix, iy, iz};
                            float lo[3], hi[3], fcell[3];
#pragma unroll
                            for (int a = 0; a < 3; a++) {
                                // cell f of the interpolation coordinate ((g + 1) / 2) (dim - 1) (the fetch's own expression)
                                const float f = floorf(((gc[a] + 1.f) / 2) * s_cell[9 + a]);
                                fcell[a] = f;
#ifdef IA_SPEC_ABL_EPS_BOX
                                lo[a] = x_l[a] - eps; hi[a] = x_l[a] + eps; (void)f;
#else
                                const float a_lo = fmaf(f, s_cell[a], s_cell[3 + a]), a_hi = fmaf(f, s_cell[a], s_cell[6 + a]);
                                const float e_lo = x_l[a] - eps, e_hi = x_l[a] + eps;
#if defined(IA_SPEC_ABL_X1)            /* ablation: the cut is computed but not used */
                                lo[a] = e_lo; hi[a] = e_hi;
                                asm volatile("" ::"v"(a_lo), "v"(a_hi));
#elif defined(IA_SPEC_ABL_X3)          /* ablation: tightness only */
                                lo[a] = e_lo; hi[a] = e_hi; (void)a_lo; (void)a_hi;
#else
                                lo[a] = a_lo > e_lo ? a_lo : e_lo;
                                hi[a] = a_hi < e_hi ? a_hi : e_hi;
#endif
#endif
                            }
#if !defined(IA_SPEC_ABL_EPS_BOX) && !defined(IA_SPEC_ABL_X1) && !defined(IA_SPEC_ABL_X2)
                            if (!(jn2 <= SPEC_TAU * SPEC_TAU)) lo[0] = INFINITY;
#endif
                            if (cell_tight) {
                                // (e) the TRUE Jacobian must be tight all over the root's cell (cell_tightness_kernel): Broyden's estimate above
                                // does not see a fold of the skinning map next to r.  The cell index is exact in float (D H W < 2^24); a valid
                                // root lies inside the grid, so its cell is an entry of the table (the last index of an axis holds 0).
                                // Bit 1: the whole 27-cell neighbourhood is tight with one orientation -- no cell cut needed (c').
                                const float ci = fmaf(fmaf(fcell[2], s_cell[10] + 1.0f, fcell[1]), s_cell[9] + 1.0f, fcell[0]);
                                const unsigned tb = cell_tight[(int)ci];
#ifndef IA_SPEC_NO_FREE3
                                if (tb & 2u) {
#pragma unroll
                                    for (int a = 0; a < 3; a++) { lo[a] = x_l[a] - eps; hi[a] = x_l[a] + eps; }
                                    if (!(jn2 <= SPEC_TAU * SPEC_TAU)) lo[0] = INFINITY;
                                }
#endif
                                if (!(tb & 1u)) lo[0] = INFINITY;
                            }
                            {   // one computed slot address instead of three predicated copies of the nine stores
                                float* const slot = rootp + n_roots * (9 * WG);
                                slot[0 * WG] = x_l[0]; slot[1 * WG] = x_l[1]; slot[2 * WG] = x_l[2];
                                slot[3 * WG] = lo[0]; slot[4 * WG] = lo[1]; slot[5 * WG] = lo[2];
                                slot[6 * WG] = hi[0]; slot[7 * WG] = hi[1]; slot[8 * WG] = hi[2];
                            }
                            n_roots++;
                        }
                    }
                }
                next = true;
                continue;
            } else if (norm_gx > dvg2) {
                if (!PACK) is_valid[index] = 0;
                next = true;
                continue;
            }
            J_inv_update(Ji, u[0], u[1], u[2], gn[0] - gx[0], gn[1] - gx[1], gn[2] - gx[2]);
            gx[0] = gn[0]; gx[1] = gn[1]; gx[2] = gn[2];
            it++;
            if (it >= 10) { if (!PACK) is_valid[index] = 0; next = true; continue; }       // not converged
        }
        // step: update = -J_inv g, x += update
        u[0] = -Ji[0] * gx[0] + -Ji[1] * gx[1] + -Ji[2] * gx[2];
        u[1] = -Ji[3] * gx[0] + -Ji[4] * gx[1] + -Ji[5] * gx[2];
        u[2] = -Ji[6] * gx[0] + -Ji[7] * gx[1] + -Ji[8] * gx[2];
        x_l[0] += u[0]; x_l[1] += u[1]; x_l[2] += u[2];
        // ---- early filter: the next fetch position against the retirement boxes of the roots later inits converged to ----
        bool near = false;
#pragma unroll
        for (int r = 0; r < SPEC_ROOTS; r++) {
            if (r < n_roots) {
                const float l0 = BOXLO(r, 0), l1 = BOXLO(r, 1), l2 = BOXLO(r, 2), h0 = BOXHI(r, 0), h1 = BOXHI(r, 1), h2 = BOXHI(r, 2);
                near = near | ((x_l[0] >= l0) & (x_l[0] < h0) & (x_l[1] >= l1) & (x_l[1] < h1) & (x_l[2] >= l2) & (x_l[2] < h2));
            }
        }
        if (near) {
            const float jn2 = fmaf(Ji[8], Ji[8], fmaf(Ji[7], Ji[7], fmaf(Ji[6], Ji[6], fmaf(Ji[5], Ji[5], fmaf(Ji[4], Ji[4],
                              fmaf(Ji[3], Ji[3], fmaf(Ji[2], Ji[2], fmaf(Ji[1], Ji[1], Ji[0] * Ji[0]))))))));
            if (jn2 <= SPEC_TAU_SELF * SPEC_TAU_SELF) { if (!PACK) is_valid[index] = 0; if (COUNT && !IA_DIAG_CELLS_ON) c_retired++; next = true; }
        }
    }
    if (COUNT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            c_fetch += __shfl_down(c_fetch, off, 64); c_retired += __shfl_down(c_retired, off, 64);
            c_valid += __shfl_down(c_valid, off, 64); c_redo += __shfl_down(c_redo, off, 64);
            c_corner += __shfl_down(c_corner, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)c_fetch);
            atomicAdd(&counters[1], (unsigned long long)c_retired);
            atomicAdd(&counters[2], (unsigned long long)c_valid);
            atomicAdd(&counters[3], (unsigned long long)c_redo);
            atomicAdd(&counters[4], (unsigned long long)c_corner);
        }
    }
}

#undef ROOT
#undef BOXLO
#undef BOXHI
#undef n_done
#undef inits
#undef flag_rec

// ---- small batches: all 13 searches of a point side by side ----------------------------------------------------------------
// broyden_spec_kernel walks the 13 inits of a point one after the other in ONE lane (that order is what lets it retire duplicates);
// a batch of fewer points than the device has lanes (the reference's 4096-ray training batches: 50 .. 200 k sample points per call)
// then takes as long as its slowest point -- ~100 dependent fetches, 0.4 ms -- however few points there are.  Here one lane owns one
// (point, init) item and runs the exact search to its end (the operation sequence of broyden_kernel, channel-last grid, x-pair loads),
// the results go straight into the flagged list -- record p = point p, all N points -- and rows_flagged_kernel applies K9 literally
// (filter.cu:10-54) and fills rows / cnt / meta / overflow records exactly as it does for the points the early filter hands over.
// Same candidates as the early-filter search wherever that search is K9-consistent (everywhere measured); by construction the
// reference's semantics.  B == 1.
__global__ __launch_bounds__(THREADS) void broyden_items_rows_kernel(
    int64_t N, int I, const float* __restrict__ xd_tgt, const float* __restrict__ voxel_J, int D, int H, int W,
    const float* __restrict__ tfs, const int32_t* __restrict__ bone_ids, const float* __restrict__ offset_g,
    const float* __restrict__ scale_g, float cvg_threshold, float dvg_threshold, float* __restrict__ J_inv, float* __restrict__ fwd_J,
    SpecFlag flag, const int32_t* __restrict__ order)
{
    const int64_t index = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (index == 0) *flag.count = (int32_t)N;
    if (index >= N * I) return;
    const int64_t p = index / I;
    const int i_init = (int)(index - p * I);
    if (i_init == 0) flag.point[p] = (int32_t)p;
    const float offset[3] = {offset_g[0], offset_g[1], offset_g[2]};
    const float scale[3] = {scale_g[0], scale_g[1], scale_g[2]};
    const int64_t src = order ? (int64_t)order[p] : p;
    float gx[3], gx_new[3] = {0, 0, 0}, xt[3], x_l[3];
    xt[0] = xd_tgt[src * 3 + 0]; xt[1] = xd_tgt[src * 3 + 1]; xt[2] = xd_tgt[src * 3 + 2];
    const float* T = tfs + (int64_t)bone_ids[i_init] * 16;
    const float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
    x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
    x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
    x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];
    float Jl[12];
    grid_sample_J<IA_LAYOUT_NDHWC, true>(voxel_J, 0, D, H, W, scale[0] * (x_l[0] + offset[0]), scale[1] * (x_l[1] + offset[1]),
                                         scale[2] * (x_l[2] + offset[2]), Jl);
    float Ji[9];
    Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
    Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
    Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];
    for (int it = 0; it < 10; it++) {
        const float J00 = Ji[0], J01 = Ji[1], J02 = Ji[2], J10 = Ji[3], J11 = Ji[4], J12 = Ji[5], J20 = Ji[6], J21 = Ji[7], J22 = Ji[8];
        if (it == 0) {
            gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3];
            gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7];
            gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11];
            gx[0] = gx[0] - xt[0]; gx[1] = gx[1] - xt[1]; gx[2] = gx[2] - xt[2];
        } else {
            gx[0] = gx_new[0]; gx[1] = gx_new[1]; gx[2] = gx_new[2];
        }
        const float u0 = -J00 * gx[0] + -J01 * gx[1] + -J02 * gx[2];
        const float u1 = -J10 * gx[0] + -J11 * gx[1] + -J12 * gx[2];
        const float u2 = -J20 * gx[0] + -J21 * gx[1] + -J22 * gx[2];
        x_l[0] += u0; x_l[1] += u1; x_l[2] += u2;
        const float ix = scale[0] * (x_l[0] + offset[0]);
        const float iy = scale[1] * (x_l[1] + offset[1]);
        const float iz = scale[2] * (x_l[2] + offset[2]);
        grid_sample_J<IA_LAYOUT_NDHWC, true>(voxel_J, 0, D, H, W, ix, iy, iz, Jl);
        gx_new[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
        gx_new[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
        gx_new[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
        const float norm_gx = gx_new[0] * gx_new[0] + gx_new[1] * gx_new[1] + gx_new[2] * gx_new[2];
        if (norm_gx < cvg_threshold * cvg_threshold) {
            const bool ok = ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1;
            if (ok) {
                float* fx = flag.x + (p * 16 + i_init) * 3;
                fx[0] = x_l[0]; fx[1] = x_l[1]; fx[2] = x_l[2];
                atomicOr(&flag.valid[p], 1u << i_init);
                if (J_inv) {
                    float* Jo = J_inv + index * 9;
                    Jo[0] = J00; Jo[1] = J01; Jo[2] = J02; Jo[3] = J10; Jo[4] = J11; Jo[5] = J12; Jo[6] = J20; Jo[7] = J21; Jo[8] = J22;
                }
                if (fwd_J) {
                    float* Fo = fwd_J + index * 9;
                    Fo[0] = Jl[0]; Fo[1] = Jl[1]; Fo[2] = Jl[2]; Fo[3] = Jl[4]; Fo[4] = Jl[5]; Fo[5] = Jl[6];
                    Fo[6] = Jl[8]; Fo[7] = Jl[9]; Fo[8] = Jl[10];
                }
            }
            return;
        } else if (norm_gx > dvg_threshold * dvg_threshold) {
            return;
        }
        J_inv_update(Ji, u0, u1, u2, gx_new[0] - gx[0], gx_new[1] - gx[1], gx_new[2] - gx[2]);
    }
}

__global__ void zero_i32_kernel(int32_t* __restrict__ p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 0; }

// ---- candidate rows of the PACK search -> packed candidate list ------------------------------------------------------------
// the points the search redid exactly: K9 as filter.cu:10-54 writes it, on all 13 results (init i is dropped when a LATER valid
// init lies within 1e-4); the survivors with the three highest inits fill the point's row, the others become overflow records
// (chained per point, lowest init at the head, keep = 1), cnt = their number
__global__ __launch_bounds__(THREADS) void rows_flagged_kernel(SpecFlag flag, int I, float* __restrict__ x_rows, int32_t* __restrict__ cnt,
                                                               uint32_t* __restrict__ meta, int32_t* __restrict__ ovf_count, int ovf_cap,
                                                               int32_t* __restrict__ ovf_head, int32_t* __restrict__ ovf_rec,
                                                               float* __restrict__ ovf_x, uint8_t* __restrict__ ovf_keep)
{
    const int n = min(*flag.count, flag.cap);
    for (int k = blockIdx.x * THREADS + threadIdx.x; k < n; k += gridDim.x * THREADS) {
        const int64_t p = flag.point[k];
        const uint32_t valid = flag.valid[k];
        const float* fx = flag.x + (int64_t)k * 16 * 3;
        int kept = 0, last = -1;
        uint32_t m = 0;
        for (int i = I - 1; i >= 0; i--) {
            if (!((valid >> i) & 1u)) continue;
            bool keep = true;
            for (int j = i + 1; j < I && keep; j++) {
                if (!((valid >> j) & 1u)) continue;
                const float d0 = fx[3 * i] - fx[3 * j], d1 = fx[3 * i + 1] - fx[3 * j + 1], d2 = fx[3 * i + 2] - fx[3 * j + 2];
                const float dist = d0 * d0 + d1 * d1 + d2 * d2;
                if ((double)dist < 0.0001 * 0.0001) keep = false;
            }
            if (!keep) continue;
            if (kept < SPEC_ROOTS) {
                float* row = x_rows + (p * SPEC_ROOTS + kept) * 3;
                row[0] = fx[3 * i]; row[1] = fx[3 * i + 1]; row[2] = fx[3 * i + 2];
                m |= (uint32_t)i << (8 * kept);
            } else {
                const int q = atomicAdd(ovf_count, 1);
                if (q < ovf_cap) {
                    ovf_rec[3 * q + 0] = (int32_t)p; ovf_rec[3 * q + 1] = i; ovf_rec[3 * q + 2] = last;
                    ovf_x[3 * q + 0] = fx[3 * i]; ovf_x[3 * q + 1] = fx[3 * i + 1]; ovf_x[3 * q + 2] = fx[3 * i + 2];
                    ovf_keep[q] = 1;
                    last = q;
                }
            }
            kept++;
        }
        cnt[p] = kept;
        meta[p] = m | (last >= 0 ? 0x80000000u : 0u);
        if (last >= 0) ovf_head[p] = last;
    }
}

// one lane per point: overflow records first (the chain from the head runs in ascending init order), then the row's candidates,
// which were stored highest init first; the packed list is in (point, ascending init) order
constexpr int FIRST_TILE_PACK = 1024;          // = FIRST_TILE (defined with its scan kernel below)
// SPLIT layout (first_pos / first_tile_off / n_first given; SDF-only queries): a point's FIRST candidate (lowest init) goes to
// fp = first_pos[p] + first_tile_off[p / 1024] -- the exclusive count of points with candidates -- and its 2nd, 3rd ... to the tail at
// n_first + (start[p] - fp) + (k - 1).  A second candidate lies
// on another body part in canonical space; kept between its neighbours' first candidates it costs the hash gather 6 % (tools/cand_order_probe.py):
// the two sub-lists are each in point order, i.e. each spatially coherent.
__global__ __launch_bounds__(THREADS) void rows_pack_kernel(int64_t N, int I, const float* __restrict__ x, const int32_t* __restrict__ cnt,
                                                             const uint32_t* __restrict__ meta, const int32_t* __restrict__ start,
                                                             const int32_t* __restrict__ ovf_head, const int32_t* __restrict__ ovf_rec,
                                                             const float* __restrict__ ovf_x, const uint8_t* __restrict__ ovf_keep,
                                                             float* __restrict__ cand_x, int32_t* __restrict__ cand_src,
                                                             const float* __restrict__ nc, const float* __restrict__ ns,
                                                             const int32_t* __restrict__ first_pos = nullptr, const int32_t* __restrict__ first_tile_off = nullptr,
                                                             const int32_t* __restrict__ n_first = nullptr)
{
    const int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= N) return;
    int c = cnt[p];
    if (c == 0) return;
    // optional: candidates leave in the hash grid's unit cube, (x - center) / scale + 0.5 (IEEE subtract, divide, add -- the bits of
    // the three elementwise passes it replaces)
    const bool nrm = nc != nullptr;
    const float c0 = nrm ? nc[0] : 0.f, c1 = nrm ? nc[1] : 0.f, c2 = nrm ? nc[2] : 0.f;
    const float s0 = nrm ? ns[0] : 1.f, s1 = nrm ? ns[1] : 1.f, s2 = nrm ? ns[2] : 1.f;
    auto put = [&](int64_t q_, float a, float b, float d) {
        if (nrm) { a = (a - c0) / s0 + 0.5f; b = (b - c1) / s1 + 0.5f; d = (d - c2) / s2 + 0.5f; }
        cand_x[q_ * 3 + 0] = a; cand_x[q_ * 3 + 1] = b; cand_x[q_ * 3 + 2] = d;
    };
    // position of the point's j-th candidate in ascending-init order
    const int64_t st = start[p];
    const int64_t fp = first_pos ? (int64_t)first_pos[p] + first_tile_off[p / FIRST_TILE_PACK] : 0, tail = first_pos ? (int64_t)*n_first + (st - fp) - 1 : 0;
    auto pos = [&](int j) -> int64_t { return first_pos ? (j == 0 ? fp : tail + j) : st + j; };
    int j = 0;
    const unsigned m = meta[p];
    if (m & 0x80000000u) {
        for (int k = ovf_head[p]; k >= 0; k = ovf_rec[3 * k + 2]) {
            if (!ovf_keep[k]) continue;
            const int64_t q = pos(j);
            put(q, ovf_x[3 * k], ovf_x[3 * k + 1], ovf_x[3 * k + 2]);
            if (cand_src) cand_src[q] = (int32_t)(p * I + ovf_rec[3 * k + 1]);
            j++;
        }
        c -= j;                                            // the rest of the point's candidates sit in its row
    }
    const float* row = x + p * (SPEC_ROOTS * 3);
    for (int k = 0; k < c; k++) {
        const int slot = c - 1 - k;
        const int64_t q = pos(j + k);
        put(q, row[slot * 3 + 0], row[slot * 3 + 1], row[slot * 3 + 2]);
        if (cand_src) cand_src[q] = (int32_t)(p * I + ((m >> (8 * slot)) & 0xffu));
    }
}

// exclusive count of "has candidates" inside tiles of FIRST_TILE points (first_local) + the tile totals; the tile offsets come from a scan of
// the (few) totals, and the consumers add them on the fly: position of p's first candidate = first_local[p] + first_tile_off[p / FIRST_TILE]
constexpr int FIRST_TILE = 1024;
__global__ __launch_bounds__(256) void first_scan_tiles_kernel(int64_t N, const int32_t* __restrict__ cnt, int32_t* __restrict__ first_local,
                                                                int32_t* __restrict__ tile_sums)
{
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * FIRST_TILE + (int64_t)threadIdx.x * 4;
    int v[4], local = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (base + k < N && cnt[base + k] > 0) ? 1 : 0; local += v[k]; }
    int inc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int run = inc - local;
    for (int w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
    for (int k = 0; k < 4; k++) { if (base + k < N) first_local[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = run;
}

// ---- K8 diagnostics ---------------------------------------------------------------
// Same search as broyden_kernel (identical arithmetic, no outputs): counts what the searches of a batch cost, for the
// L1-path figures of bench.py / DESIGN.md.  counters[0] = trilinear fetches issued, [1] = corner loads actually performed
// (in-range corners; an out-of-range corner costs no memory request), [2] = converged & in-box items, [3] = diverged items,
// [4] = items that ran out of iterations, [5 + k] = items whose search ended after k fetches (k = 2..11).
template <int LAYOUT>
__global__ __launch_bounds__(THREADS) void broyden_stats_kernel(
    int64_t total, int64_t N, int I, const float* __restrict__ xd_tgt, const float* __restrict__ voxel_J, int D, int H,
    int W, const float* __restrict__ tfs, const int32_t* __restrict__ bone_ids, const float* __restrict__ offset_g,
    const float* __restrict__ scale_g, float cvg_threshold, float dvg_threshold, unsigned long long* __restrict__ counters)
{
    const int64_t index = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    unsigned fetches = 0, corners = 0, outcome = 3;      // 0 converged, 1 diverged, 2 exhausted, 3 = no item
    if (index < total) {
        const int64_t vol = (int64_t)D * H * W;
        const int i_batch = (int)(index / (N * I));
        const int64_t i_point = (index % (N * I)) / I;
        const int i_init = (int)((index % (N * I)) % I);
        const float* vJ = (LAYOUT == IA_LAYOUT_NDHWC) ? voxel_J : voxel_J + (int64_t)i_batch * 12 * vol;
        const int vox0 = (LAYOUT == IA_LAYOUT_NDHWC) ? (int)(i_batch * vol) : 0;
        const float offset[3] = {offset_g[0], offset_g[1], offset_g[2]};
        const float scale[3] = {scale_g[0], scale_g[1], scale_g[2]};
        auto in_range_corners = [&](float gx, float gy, float gz) -> unsigned {
            float ix = ((gx + 1.f) / 2) * (W - 1), iy = ((gy + 1.f) / 2) * (H - 1), iz = ((gz + 1.f) / 2) * (D - 1);
            if (ix > 2147483646.0f || ix < -2147483648.0f || !isfinite(ix)) ix = -100.0f;
            if (iy > 2147483646.0f || iy < -2147483648.0f || !isfinite(iy)) iy = -100.0f;
            if (iz > 2147483646.0f || iz < -2147483648.0f || !isfinite(iz)) iz = -100.0f;
            const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
            const unsigned cx = (unsigned)(x0 >= 0 && x0 < W) + (unsigned)(x0 + 1 >= 0 && x0 + 1 < W);
            const unsigned cy = (unsigned)(y0 >= 0 && y0 < H) + (unsigned)(y0 + 1 >= 0 && y0 + 1 < H);
            const unsigned cz = (unsigned)(z0 >= 0 && z0 < D) + (unsigned)(z0 + 1 >= 0 && z0 + 1 < D);
            return cx * cy * cz;
        };
        float gx[3], gx_new[3] = {0, 0, 0}, xt[3], x_l[3];
        xt[0] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 0];
        xt[1] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 1];
        xt[2] = xd_tgt[((int64_t)i_batch * N + i_point) * 3 + 2];
        const float* T = tfs + ((int64_t)i_batch * 24 + bone_ids[i_init]) * 16;
        const float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
        x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
        x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
        x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];
        float Jl[12];
        {
            const float a = scale[0] * (x_l[0] + offset[0]), b = scale[1] * (x_l[1] + offset[1]), c = scale[2] * (x_l[2] + offset[2]);
            grid_sample_J<LAYOUT>(vJ, vox0, D, H, W, a, b, c, Jl);
            fetches++; corners += in_range_corners(a, b, c);
        }
        float Ji[9];
        Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
        Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
        Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];
        outcome = 2;
        for (int it = 0; it < 10; it++) {
            if (it == 0) {
                gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3];
                gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7];
                gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11];
                gx[0] = gx[0] - xt[0]; gx[1] = gx[1] - xt[1]; gx[2] = gx[2] - xt[2];
            } else {
                gx[0] = gx_new[0]; gx[1] = gx_new[1]; gx[2] = gx_new[2];
            }
            const float u0 = -Ji[0] * gx[0] + -Ji[1] * gx[1] + -Ji[2] * gx[2];
            const float u1 = -Ji[3] * gx[0] + -Ji[4] * gx[1] + -Ji[5] * gx[2];
            const float u2 = -Ji[6] * gx[0] + -Ji[7] * gx[1] + -Ji[8] * gx[2];
            x_l[0] += u0; x_l[1] += u1; x_l[2] += u2;
            const float ix = scale[0] * (x_l[0] + offset[0]);
            const float iy = scale[1] * (x_l[1] + offset[1]);
            const float iz = scale[2] * (x_l[2] + offset[2]);
            grid_sample_J<LAYOUT>(vJ, vox0, D, H, W, ix, iy, iz, Jl);
            fetches++; corners += in_range_corners(ix, iy, iz);
            gx_new[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
            gx_new[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
            gx_new[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
            const float norm_gx = gx_new[0] * gx_new[0] + gx_new[1] * gx_new[1] + gx_new[2] * gx_new[2];
            if (norm_gx < cvg_threshold * cvg_threshold) {
                outcome = (ix >= -1 && ix <= 1 && iy >= -1 && iy <= 1 && iz >= -1 && iz <= 1) ? 0 : 1;
                break;
            } else if (norm_gx > dvg_threshold * dvg_threshold) {
                outcome = 1;
                break;
            }
            J_inv_update(Ji, u0, u1, u2, gx_new[0] - gx[0], gx_new[1] - gx[1], gx_new[2] - gx[2]);
        }
    }
    // wave-level reduction, one atomic per counter and wave
    unsigned f = fetches, c = corners;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { f += __shfl_down(f, off, 64); c += __shfl_down(c, off, 64); }
    const unsigned long long m0 = __ballot(outcome == 0), m1 = __ballot(outcome == 1), m2 = __ballot(outcome == 2);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counters[0], (unsigned long long)f);
        atomicAdd(&counters[1], (unsigned long long)c);
        if (m0) atomicAdd(&counters[2], (unsigned long long)__popcll(m0));
        if (m1) atomicAdd(&counters[3], (unsigned long long)__popcll(m1));
        if (m2) atomicAdd(&counters[4], (unsigned long long)__popcll(m2));
    }
    if (outcome != 3 && fetches <= 11) atomicAdd(&counters[5 + fetches], 1ull);
}

// ---- K8 diagnostics 2: how many DISTINCT voxels do the 64 lanes of a wave fetch from? ------------------------------------
// The first two trilinear fetches of every search (47 % of all fetches; every item makes them), one item per lane, in
// point-major item order (lane l -> point l / I, init l % I: what the search kernels do) or init-major (64 consecutive points
// x one init per wave).  counters[f * 8 + k], f = fetch 0 / 1: k = 0 waves, 1 sum of distinct voxels, 2..7 histogram of the
// distinct count (1, 2, 3-4, 5-8, 9-16, 17+).  Decides whether a wave-broadcast path for the corner data can replace
// the per-lane gathers.
__global__ __launch_bounds__(THREADS) void broyden_voxel_stats_kernel(
    int64_t N, int I, int init_major, const float* __restrict__ xd_tgt, const float* __restrict__ voxel_J, int D, int H, int W,
    const float* __restrict__ tfs, const int32_t* __restrict__ bone_ids, const float* __restrict__ offset_g,
    const float* __restrict__ scale_g, unsigned long long* __restrict__ counters)
{
    const int64_t index = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t total = N * I;
    int64_t i_point;
    int i_init;
    if (init_major) {
        const int64_t blk = index / (64 * (int64_t)I);
        const int r = (int)(index - blk * 64 * I);
        i_init = r >> 6;
        i_point = blk * 64 + (r & 63);
    } else {
        i_point = index / I;
        i_init = (int)(index - i_point * I);
    }
    const bool live = index < total && i_point < N;
    const float offset[3] = {offset_g[0], offset_g[1], offset_g[2]};
    const float scale[3] = {scale_g[0], scale_g[1], scale_g[2]};
    auto voxel_of = [&](float gx, float gy, float gz) -> int {
        float ix = ((gx + 1.f) / 2) * (W - 1), iy = ((gy + 1.f) / 2) * (H - 1), iz = ((gz + 1.f) / 2) * (D - 1);
        if (ix > 2147483646.0f || ix < -2147483648.0f || !isfinite(ix)) ix = -100.0f;
        if (iy > 2147483646.0f || iy < -2147483648.0f || !isfinite(iy)) iy = -100.0f;
        if (iz > 2147483646.0f || iz < -2147483648.0f || !isfinite(iz)) iz = -100.0f;
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
        return (z0 * H + y0) * W + x0;
    };
    auto tally = [&](int fetch, int vox) {
        unsigned long long m = __ballot(live);
        const unsigned long long all = m;
        int U = 0;
        while (m) {
            const int first = __builtin_ctzll(m);
            const int v = __shfl(vox, first, 64);
            m &= ~__ballot(live && vox == v);
            U++;
        }
        if ((threadIdx.x & 63) == 0 && all) {
            const int b = U <= 1 ? 2 : U == 2 ? 3 : U <= 4 ? 4 : U <= 8 ? 5 : U <= 16 ? 6 : 7;
            atomicAdd(&counters[fetch * 8 + 0], 1ull);
            atomicAdd(&counters[fetch * 8 + 1], (unsigned long long)U);
            atomicAdd(&counters[fetch * 8 + b], 1ull);
        }
    };
    float xt[3] = {0, 0, 0}, x_l[3] = {0, 0, 0};
    float Jl[12];
    int v0 = 0, v1 = 0;
    if (live) {
        xt[0] = xd_tgt[i_point * 3 + 0]; xt[1] = xd_tgt[i_point * 3 + 1]; xt[2] = xd_tgt[i_point * 3 + 2];
        const float* T = tfs + (int64_t)bone_ids[i_init] * 16;
        const float ixd = xt[0] - T[0 * 4 + 3], iyd = xt[1] - T[1 * 4 + 3], izd = xt[2] - T[2 * 4 + 3];
        x_l[0] = ixd * T[0 * 4 + 0] + iyd * T[1 * 4 + 0] + izd * T[2 * 4 + 0];
        x_l[1] = ixd * T[0 * 4 + 1] + iyd * T[1 * 4 + 1] + izd * T[2 * 4 + 1];
        x_l[2] = ixd * T[0 * 4 + 2] + iyd * T[1 * 4 + 2] + izd * T[2 * 4 + 2];
        const float a = scale[0] * (x_l[0] + offset[0]), b = scale[1] * (x_l[1] + offset[1]), c = scale[2] * (x_l[2] + offset[2]);
        v0 = voxel_of(a, b, c);
        grid_sample_J<IA_LAYOUT_NDHWC>(voxel_J, 0, D, H, W, a, b, c, Jl);
        // J_inv starts as the transpose of the blended 3x3 block, as in the search kernels
        float Ji[9];
        Ji[0] = Jl[0]; Ji[3] = Jl[1]; Ji[6] = Jl[2];
        Ji[1] = Jl[4]; Ji[4] = Jl[5]; Ji[7] = Jl[6];
        Ji[2] = Jl[8]; Ji[5] = Jl[9]; Ji[8] = Jl[10];
        float gx[3];
        gx[0] = Jl[0] * x_l[0] + Jl[1] * x_l[1] + Jl[2] * x_l[2] + Jl[3] - xt[0];
        gx[1] = Jl[4] * x_l[0] + Jl[5] * x_l[1] + Jl[6] * x_l[2] + Jl[7] - xt[1];
        gx[2] = Jl[8] * x_l[0] + Jl[9] * x_l[1] + Jl[10] * x_l[2] + Jl[11] - xt[2];
        x_l[0] += -Ji[0] * gx[0] + -Ji[1] * gx[1] + -Ji[2] * gx[2];
        x_l[1] += -Ji[3] * gx[0] + -Ji[4] * gx[1] + -Ji[5] * gx[2];
        x_l[2] += -Ji[6] * gx[0] + -Ji[7] * gx[1] + -Ji[8] * gx[2];
        v1 = voxel_of(scale[0] * (x_l[0] + offset[0]), scale[1] * (x_l[1] + offset[1]), scale[2] * (x_l[2] + offset[2]));
    }
    tally(0, v0);
    tally(1, v1);
}

// ---- K9 -------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void filter_kernel(int64_t N, int I, const float* __restrict__ x,
                                                          const uint8_t* __restrict__ mask, uint8_t* __restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= N) return;
    for (int i = 0; i < I; i++) {
        if (!mask[p * I + i]) { out[p * I + i] = 0; continue; }
        const float xi0 = x[(p * I + i) * 3 + 0], xi1 = x[(p * I + i) * 3 + 1], xi2 = x[(p * I + i) * 3 + 2];
        bool flag = true;
        for (int j = i + 1; j < I; j++) {
            if (!mask[p * I + j]) continue;
            const float d0 = xi0 - x[(p * I + j) * 3 + 0];
            const float d1 = xi1 - x[(p * I + j) * 3 + 1];
            const float d2 = xi2 - x[(p * I + j) * 3 + 2];
            const float dist = d0 * d0 + d1 * d1 + d2 * d2;
            if ((double)dist < 0.0001 * 0.0001) { flag = false; break; }
        }
        out[p * I + i] = flag ? 1 : 0;
    }
}


// ---- per voxel cell: is the skinning map x -> A(x) x + b(x) TIGHT everywhere in the cell?  (the retirement veto of broyden_spec_kernel)
// The early filter retires a search next to a recorded root r on the promise that every search converging near r ends within K9's
// 1e-4 of it.  Broyden's J_inv at r cannot vouch for that: it starts from the bone's rigid inverse and only learns g along the steps
// the search took, so next to a FOLD of the skinning map (det dg/dx = 0: two roots 1e-4 ... 1e-3 apart in a flat valley of |g| < cvg)
// it still reads ~1.7 while the true Jacobian is near singular (tools/k9_mismatch_dump.py: every candidate-set difference left on the
// reference poses was such a pair).  The TRUE Jacobian is dg_i/dx_a = A_ia + sum_c dw_c/dx_a (A_c x + b_c)_i -- the weight-gradient term
// included -- and inside one cell it is a polynomial of the eight corner matrices: out [D,H,W] (entry = the cell whose LOW corner is the
// voxel; the last index of every axis is no cell: 0) = 1 iff at all 27 sample points of the cell (fractions 0.02 / 0.5 / 0.98 per axis)
// det has one sign and |J^-1|_F = |cof J|_F / |det J| <= tau.  A root in a cell with 0 gets no retirement box.
__global__ __launch_bounds__(THREADS) void cell_tightness_kernel(int D, int H, int W, const float* __restrict__ voxel_J_cl,
                                                                 const float* __restrict__ offset, const float* __restrict__ scale, float tau,
                                                                 uint8_t* __restrict__ out)
{
    const int64_t vox = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (vox >= (int64_t)D * H * W) return;
    const int cx = (int)(vox % W), cy = (int)((vox / W) % H), cz = (int)(vox / ((int64_t)W * H));
    if (cx >= W - 1 || cy >= H - 1 || cz >= D - 1) { out[vox] = 0; return; }
    float v[8][12];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float* src = voxel_J_cl + (vox + (c & 1) + (int64_t)((c >> 1) & 1) * W + (int64_t)((c >> 2) & 1) * W * H) * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) v[c][k] = src[k];
    }
    const int dims[3] = {W, H, D}, idx[3] = {cx, cy, cz};
    float dco[3];
#pragma unroll
    for (int a = 0; a < 3; a++) dco[a] = scale[a] * (float)(dims[a] - 1) * 0.5f;
    const float fr[3] = {0.02f, 0.5f, 0.98f};
    bool ok = true;
    float sgn = 0.0f;
    for (int s = 0; s < 27 && ok; s++) {
        const float t[3] = {fr[s % 3], fr[(s / 3) % 3], fr[s / 9]};
        float x[3];
#pragma unroll
        for (int a = 0; a < 3; a++) x[a] = (((float)idx[a] + t[a]) / (float)(dims[a] - 1) * 2.0f - 1.0f) / scale[a] - offset[a];
        float J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float wx = (c & 1) ? t[0] : 1.0f - t[0], wy = (c & 2) ? t[1] : 1.0f - t[1], wz = (c & 4) ? t[2] : 1.0f - t[2];
            const float w = wx * wy * wz;
            const float dw[3] = {((c & 1) ? 1.0f : -1.0f) * wy * wz * dco[0], ((c & 2) ? 1.0f : -1.0f) * wx * wz * dco[1],
                                 ((c & 4) ? 1.0f : -1.0f) * wx * wy * dco[2]};
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float p = v[c][4 * i] * x[0] + v[c][4 * i + 1] * x[1] + v[c][4 * i + 2] * x[2] + v[c][4 * i + 3];
#pragma unroll
                for (int a = 0; a < 3; a++) J[3 * i + a] += w * v[c][4 * i + a] + dw[a] * p;
            }
        }
        const float c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
        const float c10 = J[2] * J[7] - J[1] * J[8], c11 = J[0] * J[8] - J[2] * J[6], c12 = J[1] * J[6] - J[0] * J[7];
        const float c20 = J[1] * J[5] - J[2] * J[4], c21 = J[2] * J[3] - J[0] * J[5], c22 = J[0] * J[4] - J[1] * J[3];
        const float det = J[0] * c00 + J[1] * c01 + J[2] * c02;
        const float cof2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 + c20 * c20 + c21 * c21 + c22 * c22;
        if (s == 0) sgn = det;
        ok = (det * sgn > 0.0f) && (cof2 <= tau * tau * det * det);      // (false for NaN)
    }
    out[vox] = ok ? (uint8_t)(1 | (sgn > 0.0f ? 4 : 0)) : 0;          // bit 0: tight, bit 2: det > 0
}

// bit 1 of a tight cell's entry: all 26 neighbours are tight too, with the same sign of det -- the skinning map is steep and coherently
// oriented on the whole neighbourhood (a piecewise-smooth map with one orientation is locally injective: no second root within a cell
// width, on either side of a cell face), so a root there needs no cell cut of its retirement box.  In place: reads bits 0 / 2 of the
// neighbours, writes bit 1 of its own entry.
__global__ __launch_bounds__(THREADS) void cell_neighbourhood_kernel(int D, int H, int W, uint8_t* __restrict__ tab)
{
    const int64_t vox = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (vox >= (int64_t)D * H * W) return;
    const int cx = (int)(vox % W), cy = (int)((vox / W) % H), cz = (int)(vox / ((int64_t)W * H));
    const uint8_t me = tab[vox] & 5;
    if (!(me & 1)) return;
    if (cx < 1 || cy < 1 || cz < 1 || cx >= W - 2 || cy >= H - 2 || cz >= D - 2) return;      // a neighbour outside the grid: keep the cut
    bool ok = true;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) ok = ok && ((tab[vox + dx + (int64_t)dy * W + (int64_t)dz * W * H] & 5) == me);
    if (ok) tab[vox] = me | 2;
}

}  // namespace

IA_EXPORT int ia_precompute(int B, int D, int H, int W, const float* voxel_w, const float* tfs, const float* offset,
                            const float* scale, float* voxel_d, float* voxel_J, float* voxel_J_cl, ia_stream_t stream)
{
    IA_REQUIRE(B > 0 && D > 1 && H > 1 && W > 1, "bad grid shape");
    const int64_t vol = (int64_t)D * H * W;
    dim3 grid(ia::cdiv(vol, THREADS), B);
    precompute_kernel<<<grid, THREADS, 0, (hipStream_t)stream>>>(B, D, H, W, voxel_w, tfs, offset, scale, voxel_d,
                                                                 voxel_J, voxel_J_cl);
    return ia::check_launch("ia_precompute");
}

// uint8 [D,H,W]: 1 = the cell whose low corner is the voxel is tight (cell_tightness_kernel); the veto table of the early-filter search
IA_EXPORT int ia_cell_tightness(int D, int H, int W, const float* voxel_J_cl, const float* offset, const float* scale, float tau,
                                uint8_t* cell_tight, ia_stream_t stream)
{
    IA_REQUIRE(D > 1 && H > 1 && W > 1 && tau > 0.0f, "ia_cell_tightness: bad grid shape or tau");
    cell_tightness_kernel<<<ia::cdiv((int64_t)D * H * W, THREADS), THREADS, 0, (hipStream_t)stream>>>(D, H, W, voxel_J_cl, offset, scale, tau,
                                                                                                      cell_tight);
    int r = ia::check_launch("ia_cell_tightness");
    if (r != IA_OK) return r;
    cell_neighbourhood_kernel<<<ia::cdiv((int64_t)D * H * W, THREADS), THREADS, 0, (hipStream_t)stream>>>(D, H, W, cell_tight);
    return ia::check_launch("ia_cell_tightness(neighbourhood)");
}

IA_EXPORT int ia_fuse_broyden(int B, int64_t N, int I, const float* xd_tgt, const float* voxel_J, int layout, int D,
                              int H, int W, const float* tfs, const int32_t* bone_ids, const float* offset,
                              const float* scale, float cvg_threshold, float dvg_threshold, float* x, float* J_inv,
                              uint8_t* is_valid, float* fwd_J, ia_stream_t stream)
{
    const int64_t total = (int64_t)B * N * I;
    if (total == 0) return IA_OK;
    IA_REQUIRE(layout == IA_LAYOUT_NCDHW || layout == IA_LAYOUT_NDHWC, "unknown voxel_J layout");
    IA_REQUIRE((int64_t)B * D * H * W < ((int64_t)1 << 26), "voxel grid too large for 32-bit byte offsets (48 B per voxel)");
    hipStream_t s = (hipStream_t)stream;
    // Two bit-identical schedules.  Measured on MI355X (profiles/r01_*): primary-ray batches (most searches converge,
    // uniform length) are ~20 % faster with one item per lane; the huge secondary-ray batches (most searches diverge
    // after 1-3 steps) are ~8 % faster with the lane-persistent schedule.  Both sit near the L2-gather roofline.
    bool persistent = total >= (int64_t)32 << 20;
    if (const char* e = getenv("IA_BROYDEN_SCHEDULE")) persistent = (e[0] == 'p');     // test hook: "persistent" / "simple"
    if (persistent) {
        int br_chunk = BR_CHUNK;
        if (const char* e = getenv("IA_BR_CHUNK")) br_chunk = atoi(e);                   // tuning hook
        const int64_t n_waves = (total + br_chunk - 1) / br_chunk;
        const int grid = ia::cdiv(n_waves * 64, THREADS);
        const char* v = getenv("IA_BROYDEN_V2");
        if (B == 1 && layout == IA_LAYOUT_NDHWC && I >= 2 && I <= 16 && br_chunk < (1 << 24) && !(v && v[0] == '0')) {
            // umulhi(t, ceil(2^32 / I)) == t / I while t * (magic * I - 2^32) < 2^32, i.e. for every t < 2^28
            const uint32_t magic = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)I - 1) / (uint64_t)I);
            broyden_persistent2_kernel<<<grid, THREADS, 0, s>>>(total, I, magic, xd_tgt, voxel_J, D, H, W, tfs, bone_ids, offset, scale,
                                                                cvg_threshold, dvg_threshold, x, J_inv, is_valid, fwd_J, br_chunk);
            return ia::check_launch("ia_fuse_broyden");
        }
        if (layout == IA_LAYOUT_NDHWC)
            broyden_persistent_kernel<IA_LAYOUT_NDHWC><<<grid, THREADS, 0, s>>>(total, N, I, xd_tgt, voxel_J, D, H, W, tfs,
                                                                                bone_ids, offset, scale, cvg_threshold,
                                                                                dvg_threshold, x, J_inv, is_valid, fwd_J, br_chunk);
        else
            broyden_persistent_kernel<IA_LAYOUT_NCDHW><<<grid, THREADS, 0, s>>>(total, N, I, xd_tgt, voxel_J, D, H, W, tfs,
                                                                                bone_ids, offset, scale, cvg_threshold,
                                                                                dvg_threshold, x, J_inv, is_valid, fwd_J, br_chunk);
    } else {
        const int grid = ia::cdiv(total, THREADS);
        if (layout == IA_LAYOUT_NDHWC)
            broyden_kernel<IA_LAYOUT_NDHWC><<<grid, THREADS, 0, s>>>(total, N, I, xd_tgt, voxel_J, D, H, W, tfs, bone_ids,
                                                                     offset, scale, cvg_threshold, dvg_threshold, x, J_inv,
                                                                     is_valid, fwd_J);
        else
            broyden_kernel<IA_LAYOUT_NCDHW><<<grid, THREADS, 0, s>>>(total, N, I, xd_tgt, voxel_J, D, H, W, tfs, bone_ids,
                                                                     offset, scale, cvg_threshold, dvg_threshold, x, J_inv,
                                                                     is_valid, fwd_J);
    }
    return ia::check_launch("ia_fuse_broyden");
}

static int launch_spec(bool pack, int64_t N, int I, const float* xd_tgt, const float* voxel_J_cl, int D, int H, int W, const float* tfs,
                       const int32_t* bone_ids, const float* offset, const float* scale, float cvg_threshold, float dvg_threshold, float eps,
                       float* x, float* J_inv, uint8_t* is_valid, float* fwd_J, uint64_t* counters, int32_t* cnt, uint32_t* meta,
                       SpecFlag flag, const int32_t* order, const uint8_t* cell_tight, ia_stream_t stream, const char* what)
{
    if (N == 0) return IA_OK;
    IA_REQUIRE(I >= 1 && I <= 16, "early-filter search: 1 <= I <= 16 inits");
    IA_REQUIRE(eps >= 0.0f, "early-filter search: eps must be >= 0");
    IA_REQUIRE((int64_t)D * H * W < ((int64_t)1 << 26), "voxel grid too large for 32-bit byte offsets (48 B per voxel)");
    IA_REQUIRE(N < ((int64_t)1 << 31), "early-filter search: N must stay below 2^31");
    // chunk = 2^log2c consecutive points of the sorted order, shared by the four waves of a workgroup.  Same-box A/B on the 16.4 M headline
    // march points (tools/search_ab.py, search + rows ms): one chunk per workgroup with 2^7 / 2^8 / 2^9 / 2^10 points 10.68 / 8.81 / 9.05 /
    // 9.73 (round 3's 768: 9.41) -- ONE POINT PER LANE is best: what a short chunk loses to its slowest point it wins back through the
    // narrower window of the sorted order the resident workgroups cover (1280 x 256 points).  PERSISTENT workgroups that walk the chunks
    // b, b + G, ... as one stream (no drain at all; IA_BR_SPEC_PERSIST=1) lose badly, 13.6 - 14.7 ms: with a static stride the
    // workgroups drift apart by whole rounds and the window grows to several G x 2^log2c points -- the hardware's in-order dispatch of
    // one-chunk workgroups IS the dynamic queue that keeps the window tight.  Chunks of MORE points than lanes (the lanes that finish first
    // take the extra ones while the chunk's slow points are still running) lose as well: 288 / 320 / 384 points per 256 lanes 10.11 / 9.94 /
    // 9.70 ms against 8.90 (same box, search + rows): a second point started late in the workgroup's life extends it by a whole search.
    int log2c = 8;
    if (const char* e = getenv("IA_BR_SPEC_LOG2C")) { const int v = atoi(e); if (v >= 6 && v <= 14) log2c = v; }
    const int64_t n_chunks = (N + ((int64_t)1 << log2c) - 1) >> log2c;
    int grid = (int)n_chunks;
    if (const char* e = getenv("IA_BR_SPEC_PERSIST")) {
        if (atoi(e) == 1) {
            static int resident = 0;
            if (resident == 0) {
                int dev = 0, cus = 256, per_cu = 5;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, broyden_spec_kernel<false, true>, THREADS, 0);
                (void)hipGetLastError();
                resident = (cus > 0 ? cus : 256) * (per_cu > 0 ? per_cu : 5);
            }
            grid = (int)(n_chunks < resident ? n_chunks : resident);
        }
    }
    IA_REQUIRE(n_chunks < ((int64_t)1 << 31) && (N >> log2c) / grid < ((int64_t)1 << (30 - log2c)), "early-filter search: stream positions must fit 31 bits");
    const int pts = log2c;
    int wg_env = THREADS;
    if (const char* e = getenv("IA_BR_SPEC_WG")) { const int v = atoi(e); if (v == 64 || v == 128) wg_env = v; }
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* c = reinterpret_cast<unsigned long long*>(counters);
    int slots = SPEC_ROOTS;                                              // test hook: fewer recorded roots / row slots => more points take the exact redo
    if (const char* e = getenv("IA_SPEC_TEST_SLOTS")) { const int v = atoi(e); if (v >= 1 && v <= SPEC_ROOTS) slots = v; }
    // IA_BR_SPEC_PAD_LDS = bytes of (unused) dynamic LDS added to the launch: 31.5 KB of static LDS let five workgroups share a CU's 160 KB;
    // 8 KB more make it four (96 VGPRs x 4 waves per SIMD: 128 VGPRs per SIMD stay free for ONE wave of another stream's kernel that
    // needs no LDS -- the hash gather, 104 VGPRs).  Used with the search token of deformer.py (one search on the device at a time).
    static int pad_lds = -1;
    if (pad_lds < 0) { const char* e = getenv("IA_BR_SPEC_PAD_LDS"); const int v = e ? atoi(e) : 0; pad_lds = (v > 0 && v <= 32768) ? v : 0; }
#define IA_SPEC_LAUNCH(COUNT, PACK)                                                                                                    \
    broyden_spec_kernel<COUNT, PACK><<<grid, THREADS, pad_lds, s>>>(N, I, xd_tgt, voxel_J_cl, D, H, W, tfs, bone_ids, offset, scale,   \
                                                               cvg_threshold, dvg_threshold, eps, x, J_inv, is_valid, fwd_J, pts, c, cnt, \
                                                               meta, flag, slots, order, cell_tight)
    if (pack && !counters && wg_env != THREADS) {
        // A / B: smaller workgroups (one point per lane each): IA_BR_SPEC_WG = 64 | 128, chunk = the workgroup's lanes
        if (wg_env == 64)
            broyden_spec_kernel<false, true, 64><<<(int)((N + 63) / 64), 64, 0, s>>>(N, I, xd_tgt, voxel_J_cl, D, H, W, tfs, bone_ids, offset, scale, cvg_threshold,
                                                                                     dvg_threshold, eps, x, J_inv, is_valid, fwd_J, 6, c, cnt, meta, flag, slots, order, cell_tight);
        else
            broyden_spec_kernel<false, true, 128><<<(int)((N + 127) / 128), 128, 0, s>>>(N, I, xd_tgt, voxel_J_cl, D, H, W, tfs, bone_ids, offset, scale,
                                                                                         cvg_threshold, dvg_threshold, eps, x, J_inv, is_valid, fwd_J, 7, c, cnt, meta, flag,
                                                                                         slots, order, cell_tight);
        return ia::check_launch(what);
    }
    if (pack) { if (counters) IA_SPEC_LAUNCH(true, true); else IA_SPEC_LAUNCH(false, true); }
    else { if (counters) IA_SPEC_LAUNCH(true, false); else IA_SPEC_LAUNCH(false, false); }
#undef IA_SPEC_LAUNCH
    return ia::check_launch(what);
}

IA_EXPORT int ia_fuse_broyden_spec(int64_t N, int I, const float* xd_tgt, const float* voxel_J_cl, int D, int H, int W, const float* tfs,
                                   const int32_t* bone_ids, const float* offset, const float* scale, float cvg_threshold,
                                   float dvg_threshold, float eps, float* x, float* J_inv, uint8_t* is_valid, float* fwd_J,
                                   uint64_t* counters, const uint8_t* cell_tight, ia_stream_t stream)
{
    SpecFlag none = {nullptr, nullptr, nullptr, nullptr, 0};
    return launch_spec(false, N, I, xd_tgt, voxel_J_cl, D, H, W, tfs, bone_ids, offset, scale, cvg_threshold, dvg_threshold, eps, x, J_inv,
                       is_valid, fwd_J, counters, nullptr, nullptr, none, nullptr, cell_tight, stream, "ia_fuse_broyden_spec");
}

IA_EXPORT int ia_spec_rows_slots(void) { return SPEC_ROOTS; }

extern "C" int ia_exclusive_scan_i32(const int32_t* in, int32_t* out, int32_t* total, int64_t n, void* tmp, ia_stream_t stream);

// scratch of the rows search: [overflow count | flagged count | pad] [rec: cap x 3 int32] [x: cap x 3 float] [keep: cap bytes]
// [flagged: point (int32) | valid mask (uint32) | x (16 x 3 float)] x SPEC_FLAG_CAP
static const int SPEC_OVF_CAP = 1 << 18;
static const size_t SPEC_OVF_BYTES = 64 + (size_t)SPEC_OVF_CAP * (12 + 12 + 1) + 64;
IA_EXPORT size_t ia_spec_rows_overflow_bytes(int64_t N) { return SPEC_OVF_BYTES + (size_t)spec_flag_cap(N) * (4 + 4 + 16 * 12) + 64; }

struct OvfLayout { int32_t* count; int32_t* rec; float* x; uint8_t* keep; SpecFlag flag; };
static OvfLayout ovf_layout(void* scratch, int64_t N)
{
    const int64_t fcap = spec_flag_cap(N);
    char* b = reinterpret_cast<char*>(scratch);
    OvfLayout L;
    L.count = reinterpret_cast<int32_t*>(b);
    L.rec = reinterpret_cast<int32_t*>(b + 64);
    L.x = reinterpret_cast<float*>(b + 64 + (size_t)SPEC_OVF_CAP * 12);
    L.keep = reinterpret_cast<uint8_t*>(b + 64 + (size_t)SPEC_OVF_CAP * 24);
    char* f = b + ((SPEC_OVF_BYTES + 63) & ~(size_t)63);
    L.flag.count = reinterpret_cast<int32_t*>(b) + 1;
    L.flag.point = reinterpret_cast<int32_t*>(f);
    L.flag.valid = reinterpret_cast<uint32_t*>(f + (size_t)fcap * 4);
    L.flag.x = reinterpret_cast<float*>(f + (size_t)fcap * 8);
    L.flag.cap = (int)fcap;
    return L;
}

// total_and_overflow [1] <- max(flagged points / their capacity, overflow records / their capacity) scaled to the flagged capacity:
// anything above ia_spec_rows_overflow_capacity() means results were lost and the caller redoes the batch through is_valid + K9
__global__ void rows_report_kernel(const int32_t* __restrict__ counts, int flag_cap, int32_t* __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = counts[0] > SPEC_OVF_CAP ? flag_cap + 1 : counts[1];
}

IA_EXPORT int ia_fuse_broyden_spec_rows(int64_t N, int I, const float* xd_tgt, const float* voxel_J_cl, int D, int H, int W, const float* tfs,
                                        const int32_t* bone_ids, const float* offset, const float* scale, float cvg_threshold,
                                        float dvg_threshold, float eps, float* x_rows, float* J_inv, float* fwd_J, int32_t* cnt,
                                        uint32_t* meta, int32_t* start, int32_t* ovf_head, void* ovf_scratch, int32_t* total_and_overflow,
                                        void* scan_tmp, uint64_t* counters, const int32_t* order, const uint8_t* cell_tight, ia_stream_t stream)
{
    IA_REQUIRE(N * I < ((int64_t)1 << 31), "ia_fuse_broyden_spec_rows: N * I must stay below 2^31");
    hipStream_t s = (hipStream_t)stream;
    OvfLayout o = ovf_layout(ovf_scratch, N);
    ia::zero_bytes(o.count, 2 * sizeof(int32_t), s);
    if (N == 0) {
        ia::zero_bytes(total_and_overflow, 2 * sizeof(int32_t), s);
        return ia::check_launch("ia_fuse_broyden_spec_rows");
    }
    // small batches (IA_BR_SMALL_MAX points, default 2^18 = the flagged list's minimum capacity; 0 = off): every (point, init) search in its own lane, all points through the
    // flagged list -- see broyden_items_rows_kernel.  (counters = the fetch statistics of the early-filter kernel: that kernel runs.)
    // config-4 step, ms per call, early-filter kernel -> this path: 54 k points 0.395 -> 0.157, 115 k 0.536 -> 0.234, 180 k 0.516 -> 0.333
    int64_t small_max = (int64_t)1 << 18;                               // (read per call: the tests switch it)
    if (const char* e = getenv("IA_BR_SMALL_MAX")) { small_max = atoll(e); if (small_max > ((int64_t)1 << 18)) small_max = (int64_t)1 << 18; }
    const bool small = N <= small_max && N <= o.flag.cap && I <= 16 && counters == nullptr;
    int r;
    if (small) {
        IA_REQUIRE(I >= 1 && I <= 16, "ia_fuse_broyden_spec_rows: 1 <= I <= 16 inits");
        IA_REQUIRE((int64_t)D * H * W < ((int64_t)1 << 26), "voxel grid too large for 32-bit byte offsets (48 B per voxel)");
        ia::zero_bytes(o.flag.valid, (size_t)N * sizeof(uint32_t), s);
        broyden_items_rows_kernel<<<ia::cdiv(N * I, THREADS), THREADS, 0, s>>>(N, I, xd_tgt, voxel_J_cl, D, H, W, tfs, bone_ids, offset, scale,
                                                                              cvg_threshold, dvg_threshold, J_inv, fwd_J, o.flag, order);
        r = ia::check_launch("ia_fuse_broyden_spec_rows(items)");
    } else {
        r = launch_spec(true, N, I, xd_tgt, voxel_J_cl, D, H, W, tfs, bone_ids, offset, scale, cvg_threshold, dvg_threshold, eps, x_rows,
                        J_inv, nullptr, fwd_J, counters, cnt, meta, o.flag, order, cell_tight, stream, "ia_fuse_broyden_spec_rows");
    }
    if (r != IA_OK) return r;
    rows_flagged_kernel<<<(small ? (int)ia::cdiv(N, THREADS) : 64), THREADS, 0, s>>>(o.flag, I, x_rows, cnt, meta, o.count, SPEC_OVF_CAP, ovf_head,
                                                                                    o.rec, o.x, o.keep);
    r = ia::check_launch("ia_fuse_broyden_spec_rows(flagged)");
    if (r != IA_OK) return r;
    if (small) zero_i32_kernel<<<1, 64, 0, s>>>(o.flag.count);        // "points redone exactly" of the report below: none, by design
    r = ia_exclusive_scan_i32(cnt, start, total_and_overflow, N, scan_tmp, stream);
    if (r != IA_OK) return r;
    // [1] = number of points redone exactly; the caller compares it with ia_spec_rows_overflow_capacity()
    rows_report_kernel<<<1, 64, 0, s>>>(o.count, o.flag.cap, total_and_overflow + 1);
    return ia::check_launch("ia_fuse_broyden_spec_rows(report)");
}

IA_EXPORT int64_t ia_spec_rows_overflow_capacity(int64_t N) { return spec_flag_cap(N); }

IA_EXPORT int ia_deform_rows_pack(int64_t N, int I, const float* x_rows, const int32_t* cnt, const uint32_t* meta, const int32_t* start,
                                  const int32_t* ovf_head, const void* ovf_scratch, float* cand_x, int32_t* cand_src,
                                  const float* norm_center, const float* norm_scale, ia_stream_t stream)
{
    if (N == 0) return IA_OK;
    IA_REQUIRE(cand_x != x_rows, "ia_deform_rows_pack: cand_x must not alias x_rows");
    IA_REQUIRE((norm_center == nullptr) == (norm_scale == nullptr), "ia_deform_rows_pack: norm_center and norm_scale go together");
    OvfLayout o = ovf_layout(const_cast<void*>(ovf_scratch), N);
    rows_pack_kernel<<<ia::cdiv(N, THREADS), THREADS, 0, (hipStream_t)stream>>>(N, I, x_rows, cnt, meta, start, ovf_head, o.rec, o.x, o.keep,
                                                                                cand_x, cand_src, norm_center, norm_scale);
    return ia::check_launch("ia_deform_rows_pack");
}

// ia_deform_rows_pack in the SPLIT layout (rows_pack_kernel).  Outputs for the reader (ia_deform_select_min_split): first_pos [N] int32 =
// exclusive count of points that have candidates INSIDE the point's tile of 1024, first_tile_off [ceil(N / 1024)] int32 = the tiles' offsets,
// n_first [1] int32 (DEVICE) = the number of such points; the first candidate of p sits at first_pos[p] + first_tile_off[p / 1024], its other
// candidates from n_first on, point-major.  scan_tmp: ia_scan_tmp_bytes(N / 1024 + 1) + 4 (N / 1024 + 1) + 512 bytes.
IA_EXPORT int ia_deform_rows_pack_split(int64_t N, int I, const float* x_rows, const int32_t* cnt, const uint32_t* meta, const int32_t* start,
                                        const int32_t* ovf_head, const void* ovf_scratch, int32_t* first_pos, int32_t* first_tile_off,
                                        int32_t* n_first, float* cand_x, const float* norm_center, const float* norm_scale, void* scan_tmp,
                                        ia_stream_t stream)
{
    if (N == 0) return IA_OK;
    static_assert(FIRST_TILE == FIRST_TILE_PACK, "one tile size");
    IA_REQUIRE(cand_x != x_rows, "ia_deform_rows_pack_split: cand_x must not alias x_rows");
    IA_REQUIRE((norm_center == nullptr) == (norm_scale == nullptr), "ia_deform_rows_pack_split: norm_center and norm_scale go together");
    IA_REQUIRE(first_pos != nullptr && first_tile_off != nullptr && n_first != nullptr && scan_tmp != nullptr,
               "ia_deform_rows_pack_split: first_pos, first_tile_off, n_first and scan_tmp are required");
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = (N + FIRST_TILE - 1) / FIRST_TILE;
    int32_t* sums = reinterpret_cast<int32_t*>(scan_tmp);
    void* tmp = reinterpret_cast<char*>(scan_tmp) + (((size_t)tiles * 4 + 255) & ~(size_t)255);
    first_scan_tiles_kernel<<<(int)tiles, 256, 0, s>>>(N, cnt, first_pos, sums);
    const int r = ia_exclusive_scan_i32(sums, first_tile_off, n_first, tiles, tmp, stream);
    if (r != IA_OK) return r;
    OvfLayout o = ovf_layout(const_cast<void*>(ovf_scratch), N);
    rows_pack_kernel<<<ia::cdiv(N, THREADS), THREADS, 0, s>>>(N, I, x_rows, cnt, meta, start, ovf_head, o.rec, o.x, o.keep, cand_x, nullptr,
                                                              norm_center, norm_scale, first_pos, first_tile_off, n_first);
    return ia::check_launch("ia_deform_rows_pack_split");
}

IA_EXPORT int ia_filter(int64_t N, int I, const float* x, const uint8_t* mask, uint8_t* out, ia_stream_t stream)
{
    if (N == 0) return IA_OK;
    filter_kernel<<<ia::cdiv(N, THREADS), THREADS, 0, (hipStream_t)stream>>>(N, I, x, mask, out);
    return ia::check_launch("ia_filter");
}

IA_EXPORT int ia_broyden_stats(int B, int64_t N, int I, const float* xd_tgt, const float* voxel_J, int layout, int D, int H,
                               int W, const float* tfs, const int32_t* bone_ids, const float* offset, const float* scale,
                               float cvg_threshold, float dvg_threshold, uint64_t* counters /*[17], caller-zeroed, accumulated*/,
                               ia_stream_t stream)
{
    const int64_t total = (int64_t)B * N * I;
    if (total == 0) return IA_OK;
    IA_REQUIRE(layout == IA_LAYOUT_NCDHW || layout == IA_LAYOUT_NDHWC, "unknown voxel_J layout");
    IA_REQUIRE((int64_t)B * D * H * W < ((int64_t)1 << 26), "voxel grid too large for 32-bit byte offsets");
    const int grid = ia::cdiv(total, THREADS);
    unsigned long long* c = reinterpret_cast<unsigned long long*>(counters);
    if (layout == IA_LAYOUT_NDHWC)
        broyden_stats_kernel<IA_LAYOUT_NDHWC><<<grid, THREADS, 0, (hipStream_t)stream>>>(
            total, N, I, xd_tgt, voxel_J, D, H, W, tfs, bone_ids, offset, scale, cvg_threshold, dvg_threshold, c);
    else
        broyden_stats_kernel<IA_LAYOUT_NCDHW><<<grid, THREADS, 0, (hipStream_t)stream>>>(
            total, N, I, xd_tgt, voxel_J, D, H, W, tfs, bone_ids, offset, scale, cvg_threshold, dvg_threshold, c);
    return ia::check_launch("ia_broyden_stats");
}

// diagnostics: distinct voxels per wave for the first two fetches of every search (see broyden_voxel_stats_kernel);
// counters [16], caller-zeroed.  B = 1, channel-last grid.
IA_EXPORT int ia_broyden_voxel_stats(int64_t N, int I, int init_major, const float* xd_tgt, const float* voxel_J, int D, int H, int W,
                                     const float* tfs, const int32_t* bone_ids, const float* offset, const float* scale,
                                     uint64_t* counters, ia_stream_t stream)
{
    if (N == 0) return IA_OK;
    const int64_t total = ((N + 63) / 64) * 64 * I;
    broyden_voxel_stats_kernel<<<ia::cdiv(total, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        N, I, init_major, xd_tgt, voxel_J, D, H, W, tfs, bone_ids, offset, scale, reinterpret_cast<unsigned long long*>(counters));
    return ia::check_launch("ia_broyden_voxel_stats");
}
