// mlp.hip -- fused small-MLP evaluation on the CDNA4 matrix cores (fp32-in / fp32-acc MFMA).
// Replaces the PyTorch MLPs of the reference's field queries:
//   VanillaMLP  35->64->13  (Softplus beta=100, weight-norm)  models/network_utils.py:201-244  (SDF, rf/geometry.py:152)
//   VanillaMLP  67->64->64->3 (ReLU)                          models/rf/radiance.py:118-131
//   LipshitzMLP 48->64->64->5 (ReLU)                          models/network_utils.py:360-431, pbr/material.py:31-51
// and, for the SDF head, the analytic normal that the reference obtains with
// torch.autograd.grad(create_graph=True) (rf/geometry.py:165-172).
//
// Design
//   * one wave owns a tile of 32 points; its activations never leave LDS between layers.  Small tiles on purpose:
//     a 32 x 69 float tile is 8.8 KB, so 12 waves (3 per SIMD) fit next to the weights in one CU's LDS and the
//     global-load latency of one wave's input assembly hides behind the MFMA chains of the other two
//     (the 64-point / 1-wave-per-SIMD version of this kernel sat at ~20 % of the MFMA-bound time);
//   * the input row is ASSEMBLED in LDS from up to 5 source segments (hash features, xyz, geometry
//     feature, SH, normal ...) so the concatenated [n, 67] tensor is never materialised in HBM;
//   * hidden layers (N = 64): v_mfma_f32_32x32x2_f32, 1x2 tiles of 32x32 per wave;
//     output layer (N <= 16): v_mfma_f32_16x16x4_f32 (2 tiles of 16 points);
//     fp32 MFMA is bit-equal to an fmaf chain, so parity mode needs no reduced precision;
//   * weights (effective: weight-norm / Lipschitz scaling / level masks folded in on the host) are
//     staged once per workgroup in LDS, rows padded to an odd stride => conflict-free ds_read_b32
//     for both operand patterns;
//   * SDF head: g_z = softplus'(z) * W2[0,:], g_h = g_z W1 (one more MFMA GEMM), then a per-point
//     epilogue contracts g_h with the hash-grid Jacobian (d enc / d x) -> analytic gradient.
#include <cstring>
#include <type_traits>
#include <stdlib.h>

#include "mlp_tile.h"

namespace {

using mlp::Seg;
using mlp::MAX_SEGS;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 32;                  // points per wave tile
constexpr int MT = TM / 32;
constexpr int HID = 64;
struct MlpArgs {
    int64_t n;
    int n_segs;
    Seg segs[MAX_SEGS];
    const float *W1, *b1, *W2, *b2, *Wo, *bo;    // W1 [64,IN], W2 [64,64] (NHID==2), Wo [OUT,64]
    float* y;                                    // [n, y_stride] (first OUT columns written)
    int y_stride;
    // SDF head extras
    const float* jac;        // [n, 32, 3] d enc / d x'
    int xyz_col;             // column of the first xyz input in the assembled row
    float inv_scale[3];      // 1 / bbox extent
    float* grad;             // [n, 3]
};

__device__ __forceinline__ float act_hidden(float v, int hact)
{
    if (hact == 0) return fmaxf(v, 0.0f);
    const float bx = 100.0f * v;                      // Softplus(beta=100, threshold=20)
    return bx > 20.0f ? v : log1pf(__expf(bx)) * 0.01f;
}

template <int KIND, int IN, int NHID, int OUT, int HACT, int OACT, bool SDF_GRAD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mlp_fwd_kernel(MlpArgs a)
{
    constexpr int THREADS = WAVES * 64;
    constexpr int IN_PAD = (IN + 1) / 2 * 2;
    constexpr int LDW1 = IN_PAD + 1;
    constexpr int LDW = HID + 1;
    constexpr int LDX = (IN_PAD > HID ? IN_PAD : HID) + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;                                   // [64][LDW1]
    float* sW2 = sW1 + HID * LDW1;                       // [64][LDW]   (NHID == 2)
    float* sWo = sW2 + (NHID == 2 ? HID * LDW : 0);      // [16][LDW]
    float* sB = sWo + 16 * LDW;                          // b1[64] b2[64] bo[16]
    float* sXall = sB + 64 + 64 + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sX = sXall + wave * TM * LDX;                 // this wave's [TM][LDX] tile
    float* sG = sX;                                      // SDF_GRAD: reused for g_z / g_h

    // ---- stage weights (zero padded) ----
    for (int i = tid; i < HID * LDW1; i += THREADS) {
        const int r = i / LDW1, c = i % LDW1;
        sW1[i] = (c < IN) ? a.W1[r * IN + c] : 0.0f;
    }
    if (NHID == 2)
        for (int i = tid; i < HID * LDW; i += THREADS) {
            const int r = i / LDW, c = i % LDW;
            sW2[i] = (c < HID) ? a.W2[r * HID + c] : 0.0f;
        }
    for (int i = tid; i < 16 * LDW; i += THREADS) {
        const int r = i / LDW, c = i % LDW;
        sWo[i] = (r < OUT && c < HID) ? a.Wo[r * HID + c] : 0.0f;
    }
    if (tid < 64) { sB[tid] = a.b1[tid]; sB[64 + tid] = (NHID == 2) ? a.b2[tid] : 0.0f; }
    if (tid < 16) sB[128 + tid] = (tid < OUT) ? a.bo[tid] : 0.0f;
    __syncthreads();

    const int64_t n_tiles = (a.n + TM - 1) / TM;
    // (tried: register prefetch of the next tile's segments -- at 12 waves / CU it spills (168-register budget), at
    //  8 waves / CU it loses more to the missing third wave than it gains: 0.48 / 0.46 vs 0.50 / 0.56 MFMA utilisation)
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t p0 = tile * TM;
        // ---- assemble the input rows in LDS (all global loads of the tile in flight together) ----
        mlp::assemble<KIND, IN, TM>(sX, LDX, a.segs, p0, a.n, lane);

        const int lr = lane & 31, lk = lane >> 5;
        // ---- layer 1: [64 x IN_PAD] x W1^T -> [64 x 64] ----
        f32x16 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[m][nt][r] = 0.0f;
        {   // software-pipelined: the operands of k-step kk+1 are fetched from LDS while the MFMAs of kk run
            float b0n = sW1[lr * LDW1 + lk], b1n = sW1[(32 + lr) * LDW1 + lk], an[MT];
#pragma unroll
            for (int m = 0; m < MT; m++) an[m] = sX[(32 * m + lr) * LDX + lk];
#pragma unroll 2
            for (int kk = 0; kk < IN_PAD / 2; kk++) {
                const float b0 = b0n, b1v = b1n;
                float av[MT];
#pragma unroll
                for (int m = 0; m < MT; m++) av[m] = an[m];
                if (kk + 1 < IN_PAD / 2) {
                    const int k = 2 * (kk + 1) + lk;
                    b0n = sW1[lr * LDW1 + k]; b1n = sW1[(32 + lr) * LDW1 + k];
#pragma unroll
                    for (int m = 0; m < MT; m++) an[m] = sX[(32 * m + lr) * LDX + k];
                }
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b0, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b1v, acc[m][1], 0, 0, 0);
                }
            }
        }
        // epilogue: bias + activation -> sX (in place: all reads of the old tile are complete)
        // SDF_GRAD keeps z in registers to form softplus'(z) later
        float sig[MT][2][16];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lk, col = 32 * nt + lr;
                    const float z = acc[m][nt][r] + sB[col];
                    if (HACT == 1) {
                        float sg;
                        sX[row * LDX + col] = mlp::softplus100(z, sg);
                        if (SDF_GRAD) sig[m][nt][r] = sg;
                    } else {
                        sX[row * LDX + col] = fmaxf(z, 0.0f);
                    }
                }
        // ---- layer 2 (optional) ----
        if (NHID == 2) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[m][nt][r] = 0.0f;
            {
                float b0n = sW2[lr * LDW + lk], b1n = sW2[(32 + lr) * LDW + lk], an[MT];
#pragma unroll
                for (int m = 0; m < MT; m++) an[m] = sX[(32 * m + lr) * LDX + lk];
#pragma unroll 2
                for (int kk = 0; kk < HID / 2; kk++) {
                    const float b0 = b0n, b1v = b1n;
                    float av[MT];
#pragma unroll
                    for (int m = 0; m < MT; m++) av[m] = an[m];
                    if (kk + 1 < HID / 2) {
                        const int k = 2 * (kk + 1) + lk;
                        b0n = sW2[lr * LDW + k]; b1n = sW2[(32 + lr) * LDW + k];
#pragma unroll
                        for (int m = 0; m < MT; m++) an[m] = sX[(32 * m + lr) * LDX + k];
                    }
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b0, acc[m][0], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b1v, acc[m][1], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lk, col = 32 * nt + lr;
                        sX[row * LDX + col] = act_hidden(acc[m][nt][r] + sB[64 + col], HACT);
                    }
        }
        // ---- output layer: [64 x 64] x Wo^T -> [64 x 16] with 16x16x4 tiles ----
        {
            const int l15 = lane & 15, l4 = lane >> 4;
            constexpr int OT = TM / 16;
            f32x4 o[OT];
#pragma unroll
            for (int m = 0; m < OT; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) o[m][r] = 0.0f;
            float bn = sWo[l15 * LDW + l4], xn[OT];
#pragma unroll
            for (int m = 0; m < OT; m++) xn[m] = sX[(16 * m + l15) * LDX + l4];
#pragma unroll 4
            for (int kk = 0; kk < HID / 4; kk++) {
                const float b = bn;
                float xv[OT];
#pragma unroll
                for (int m = 0; m < OT; m++) xv[m] = xn[m];
                if (kk + 1 < HID / 4) {
                    const int k = 4 * (kk + 1) + l4;
                    bn = sWo[l15 * LDW + k];
#pragma unroll
                    for (int m = 0; m < OT; m++) xn[m] = sX[(16 * m + l15) * LDX + k];
                }
#pragma unroll
                for (int m = 0; m < OT; m++) o[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[m], b, o[m], 0, 0, 0);
            }
            if (l15 < OUT) {
                const float bias = sB[128 + l15];
#pragma unroll
                for (int m = 0; m < OT; m++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int64_t p = p0 + 16 * m + l4 * 4 + r;
                        float v = o[m][r] + bias;
                        if (OACT == 1) v = 1.0f / (1.0f + __expf(-v));
                        if (p < a.n) a.y[p * a.y_stride + l15] = v;
                    }
            }
        }
        // ---- SDF head: analytic gradient ----
        if (SDF_GRAD) {
            // g_z[p][o] = softplus'(z[p][o]) * Wo[0][o]  -> sG (over the hidden activations, now dead)
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lk, col = 32 * nt + lr;
                        sG[row * LDX + col] = sig[m][nt][r] * sWo[col];
                    }
            // g_h = g_z W1 : A = g_z [64 pts x 64], B[k][j] = W1[k][j], j < IN_PAD (<= 64)
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[m][nt][r] = 0.0f;
            {
                const bool c0ok = lr < IN_PAD, c1ok = 32 + lr < IN_PAD;
                float b0n = c0ok ? sW1[lk * LDW1 + lr] : 0.0f, b1n = c1ok ? sW1[lk * LDW1 + 32 + lr] : 0.0f, an[MT];
#pragma unroll
                for (int m = 0; m < MT; m++) an[m] = sG[(32 * m + lr) * LDX + lk];
#pragma unroll 2
                for (int kk = 0; kk < HID / 2; kk++) {
                    const float b0 = b0n, b1v = b1n;
                    float av[MT];
#pragma unroll
                    for (int m = 0; m < MT; m++) av[m] = an[m];
                    if (kk + 1 < HID / 2) {
                        const int k = 2 * (kk + 1) + lk;
                        b0n = c0ok ? sW1[k * LDW1 + lr] : 0.0f;
                        b1n = c1ok ? sW1[k * LDW1 + 32 + lr] : 0.0f;
#pragma unroll
                        for (int m = 0; m < MT; m++) an[m] = sG[(32 * m + lr) * LDX + k];
                    }
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b0, acc[m][0], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b1v, acc[m][1], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lk, col = 32 * nt + lr;
                        sG[row * LDX + col] = acc[m][nt][r];
                    }
            // per-point epilogue (lane = point): grad_a = (2 g_h[xyz_a] + sum_k g_h[k] J[k][a]) * inv_scale_a
            const int64_t p = p0 + lane;
            if (lane < TM && p < a.n) {
                const float* gh = sG + lane * LDX;
                float g0 = 2.0f * gh[a.xyz_col + 0], g1 = 2.0f * gh[a.xyz_col + 1], g2 = 2.0f * gh[a.xyz_col + 2];
                if constexpr (KIND == 3) {
                    // level-major Jacobian [16][n][6] (as the XCD-partitioned gather leaves it): consecutive lanes = consecutive
                    // points read consecutive 24-byte records of a level -- coalesced, where the [n,32,3] rows are 384 bytes apart
                    const float* J = a.jac + p * 6;
                    const int64_t ls = a.n * 6;
#pragma unroll 4
                    for (int l = 0; l < 16; l++) {
                        const float2 ja = *reinterpret_cast<const float2*>(J + l * ls), jb = *reinterpret_cast<const float2*>(J + l * ls + 2),
                                     jc = *reinterpret_cast<const float2*>(J + l * ls + 4);
                        const float ga = gh[2 * l], gb = gh[2 * l + 1];      // same order of operations as the row-major loop
                        g0 = fmaf(ga, ja.x, g0); g1 = fmaf(ga, ja.y, g1); g2 = fmaf(ga, jb.x, g2);
                        g0 = fmaf(gb, jb.y, g0); g1 = fmaf(gb, jc.x, g1); g2 = fmaf(gb, jc.y, g2);
                    }
                } else {
                const float* J = a.jac + p * 96;
#pragma unroll 8
                for (int k = 0; k < 32; k++) {
                    const float g = gh[k];      // hash features occupy columns 0..31
                    g0 = fmaf(g, J[k * 3 + 0], g0);
                    g1 = fmaf(g, J[k * 3 + 1], g1);
                    g2 = fmaf(g, J[k * 3 + 2], g2);
                }
                }
                a.grad[p * 3 + 0] = g0 * a.inv_scale[0];
                a.grad[p * 3 + 1] = g1 * a.inv_scale[1];
                a.grad[p * 3 + 2] = g2 * a.inv_scale[2];
            }
        }
    }
}

template <int KIND, int IN, int NHID, int OUT, int HACT, int OACT, bool SDF_GRAD>
int launch_fwd(const MlpArgs& a, hipStream_t s)
{
    constexpr int WAVES = 12;                 // 3 waves per SIMD, one workgroup per CU
    constexpr int IN_PAD = (IN + 1) / 2 * 2;
    constexpr int LDW1 = IN_PAD + 1, LDW = HID + 1, LDX = (IN_PAD > HID ? IN_PAD : HID) + 1;
    constexpr size_t lds = sizeof(float) * (HID * LDW1 + (NHID == 2 ? HID * LDW : 0) + 16 * LDW + 144 + WAVES * TM * LDX);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = mlp_fwd_kernel<KIND, IN, NHID, OUT, HACT, OACT, SDF_GRAD, WAVES>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int64_t n_tiles = (a.n + TM - 1) / TM;
    int grid = (int)((n_tiles + WAVES - 1) / WAVES);
    if (grid > 256) grid = 256;               // persistent: one workgroup per CU, tiles strided over the grid
    kern<<<grid, WAVES * 64, lds, s>>>(a);
    return ia::check_launch("ia_mlp_fwd");
}


// ---- SDF value head, software-pipelined (ia_sdf_levels_fwd) ---------------------------------------------------------------
// 35 -> 64 (Softplus beta=100) -> 1 on the level-major hash features.  mlp_fwd_kernel<3, ..., OUT = 1> spends a 32-point tile
// as  [layer-1 MFMAs: 36 x 64 clk on the matrix pipe] THEN [Softplus + operand staging: ~1200 VALU instructions] THEN
// [a 16-wide output-layer MFMA tile for ONE output column]; counters showed the VALU pipes 57 % + the matrix pipe 40 % busy =
// 97 % of the launch -- the two halves ran back to back, never together (profiles/r02: 0.30-0.35 of the fp32 MFMA peak).
// Here ONE wave carries TWO tiles in flight: the layer-1 MFMAs of tile t+1 are issued in the SAME instruction stream as the
// Softplus of tile t (independent registers: the MFMA executes on the matrix pipe while the wave goes on issuing VALU work;
// sched_group_barrier pins the 1 MFMA : ~14 VALU interleave), the output layer is a dot product done where the activations are
// (lane-partial FMAs + one LDS transpose-reduce per tile: no 16x16x4 tile for one column, no activation round trip through
// LDS), and the global loads of tile t+2 are in flight behind both.
constexpr int SH_LDX = 37;              // 36 input columns (32 hash features | xyz | pad) + 1: odd row stride
constexpr int SH_LDW = 37;

struct ShRows { float2 q[8]; float v[2]; };

__device__ __forceinline__ void sh_load(ShRows& r, const float2* __restrict__ lv, int64_t ls, const float* __restrict__ xp, int64_t p0,
                                        int64_t n, int lane)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = j * 64 + lane, lvl = i >> 5, row = i & 31;
        r.q[j] = (p0 + row < n) ? lv[(int64_t)lvl * ls + p0 + row] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {                                     // 32 rows x 3 xyz = 96 scalars
        const int i = j * 64 + lane;
        const int64_t p = p0 + i / 3;
        r.v[j] = (i < 96 && p < n) ? xp[p0 * 3 + i] : 0.5f;
    }
}

__device__ __forceinline__ void sh_store(const ShRows& r, float* __restrict__ sT, int64_t p0, int64_t n, int lane)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = j * 64 + lane, lvl = i >> 5, row = i & 31;
        sT[row * SH_LDX + 2 * lvl] = r.q[j].x; sT[row * SH_LDX + 2 * lvl + 1] = r.q[j].y;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int i = j * 64 + lane;
        if (i < 96) { const int row = i / 3, c = i - row * 3; sT[row * SH_LDX + 32 + c] = (p0 + row < n) ? r.v[j] * 2.0f - 1.0f : 0.0f; }
    }
    if (lane < 32) sT[lane * SH_LDX + 35] = 0.0f;
}

// value-only Softplus(beta = 100) times an output weight, straight-line: 6 full-rate instructions + exp2 + log2 (56 clk per
// wave-activation, under the 64 clk of the 32x32x2 MFMA it hides behind).  b100 = 100 * bias, w001 = 0.01 * weight.
//   softplus(z) = (max(bx, 0) + log(1 + exp(-|bx|))) / 100,  bx = 100 z.
// Against torch.nn.Softplus(beta=100, threshold=20) (models/network_utils.py:240): no pass-through above the threshold (there
// exp(-bx) < 2.1e-9 and the formula returns z to within 2 ulp), and log1p as log(1 + e) (absolute error <= 6e-8 in the
// logarithm = 6e-10 in the activation; the SDF sums 64 of them with weights O(1): far inside the head's 5e-6 tolerance).
__device__ __forceinline__ float softplus100_times(float acc, float b100, float w001, float part)
{
    const float bx = fmaf(acc, 100.0f, b100);
    const float e = __builtin_amdgcn_exp2f(-fabsf(bx) * 1.44269504088896340736f);      // exp(-|bx|) in (0, 1]
    const float lg2 = __builtin_amdgcn_logf(1.0f + e);                                  // log2(1 + e)
    return fmaf(fmaf(lg2, 0.69314718055994530942f, fmaxf(bx, 0.0f)), w001, part);
}

template <int SH_WAVES, bool PROBE_CACHED = false>     // PROBE_CACHED (diagnostic, IA_SDF_HEAD=probe_cached): every tile reads the rows of one of 64 tiles
__global__ __launch_bounds__(SH_WAVES * 64) void sdf_head_pipelined_kernel(int64_t n, const float2* __restrict__ levels,
                                                                           const float* __restrict__ xp, const float* __restrict__ W1,
                                                                           const float* __restrict__ b1, const float* __restrict__ Wo,
                                                                           const float* __restrict__ bo, float* __restrict__ sdf)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;                                           // [64][SH_LDW], columns 35 (pad) zero
    float* sXall = sW1 + 64 * SH_LDW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const sX0 = sXall + wave * 2 * 32 * SH_LDX;            // this wave's two row buffers: sX0 + b * 32 * SH_LDX
    for (int i = tid; i < 64 * SH_LDW; i += SH_WAVES * 64) {
        const int r = i / SH_LDW, c = i % SH_LDW;
        sW1[i] = (c < 35) ? W1[r * 35 + c] : 0.0f;
    }
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    const float w0 = 0.01f * Wo[lr], w1 = 0.01f * Wo[32 + lr], bb0 = 100.0f * b1[lr], bb1 = 100.0f * b1[32 + lr], bout = bo[0];
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * SH_WAVES;
    int64_t tile = (int64_t)blockIdx.x * SH_WAVES + wave;
    if (tile >= n_tiles) return;

    f32x16 accA0, accA1, accB0, accB1;
    ShRows rows;
    // prologue: tile 0 of this wave -> LDS buffer 0 -> layer 1 -> accA; tile 1 -> LDS buffer 1; tile 2 -> registers
    auto src = [](int64_t t) -> int64_t { return PROBE_CACHED ? (t & 63) : t; };
    sh_load(rows, levels, n, xp, src(tile) * 32, n, lane);
    sh_store(rows, sX0, tile * 32, n, lane);
    if (tile + stride < n_tiles) sh_load(rows, levels, n, xp, src(tile + stride) * 32, n, lane);
#pragma unroll
    for (int r = 0; r < 16; r++) { accA0[r] = 0.f; accA1[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 18; kk++) {
        const int k = 2 * kk + lk;
        const float a = sX0[lr * SH_LDX + k], b0 = sW1[lr * SH_LDW + k], b1v = sW1[(32 + lr) * SH_LDW + k];
        accA0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, accA0, 0, 0, 0);
        accA1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1v, accA1, 0, 0, 0);
    }
    if (tile + stride < n_tiles) sh_store(rows, sX0 + 32 * SH_LDX, (tile + stride) * 32, n, lane);
    int cur = 0;                                                  // LDS buffer holding the CONSUMED rows of `tile` (free for the reduce)
    for (;; tile += stride) {
        const bool have_next = tile + stride < n_tiles;           // its rows are in sXb[cur ^ 1]
        const bool have_next2 = tile + 2 * stride < n_tiles;
        if (have_next2) sh_load(rows, levels, n, xp, src(tile + 2 * stride) * 32, n, lane);      // in flight behind everything below
        float part[16];
        const float* sN = sX0 + (cur ^ 1) * 32 * SH_LDX;
#pragma unroll
        for (int r = 0; r < 16; r++) { accB0[r] = 0.f; accB1[r] = 0.f; }
        // (a wave whose last tile has no successor runs the MFMAs below on stale rows: one wasted layer-1 per wave, no second code path)
        {
        float an = sN[lr * SH_LDX + lk], b0n = sW1[lr * SH_LDW + lk], b1n = sW1[(32 + lr) * SH_LDW + lk];
#pragma unroll
        for (int kk = 0; kk < 18; kk++) {
            const float a = an, b0 = b0n, b1v = b1n;
            if (kk + 1 < 18) {
                const int k = 2 * (kk + 1) + lk;
                an = sN[lr * SH_LDX + k]; b0n = sW1[lr * SH_LDW + k]; b1n = sW1[(32 + lr) * SH_LDW + k];
            }
            accB0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, accB0, 0, 0, 0);
            if (kk < 16) part[kk] = softplus100_times(accA0[kk], bb0, w0, 0.0f);       // Softplus of tile t under the MFMAs of tile t+1
            accB1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1v, accB1, 0, 0, 0);
            if (kk < 16) part[kk] = softplus100_times(accA1[kk], bb1, w1, part[kk]);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);       // one activation of VALU work
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);       // the next k-step's operands
        }
        }
        // output layer: out[row] = bo + sum over the 64 hidden units; this lane holds the partial of (row(r, lk), columns lr, 32 + lr)
        float* sR = sX0 + cur * 32 * SH_LDX;                      // [32 rows][32 partials], row stride SH_LDX
#pragma unroll
        for (int r = 0; r < 16; r++) sR[((r & 3) + 8 * (r >> 2) + 4 * lk) * SH_LDX + lr] = part[r];
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) sum += sR[lr * SH_LDX + lk * 16 + j];
        sum += __shfl_xor(sum, 32, 64);
        const int64_t p = tile * 32 + lr;
        if (lk == 0 && p < n) sdf[p] = sum + bout;
        if (!have_next) break;
        if (have_next2) sh_store(rows, sR, (tile + 2 * stride) * 32, n, lane);          // over the reduce scratch: same wave, in order
        accA0 = accB0; accA1 = accB1;
        cur ^= 1;
    }
}

// ---- second version: the same pipeline with fewer issue slots per tile ----------------------------------------------------
// Counters on the kernel above (tools/pmc_probe.sh, 50 M points): per 32-point tile and wave 36 MFMA (64 clk each on the matrix
// pipe: busy 50 % of the launch) against 410 other VALU, 79 LDS and ~120 scalar instructions; the VALU issue port is busy 44 %, the
// waves spend 65 % of their cycles waiting to issue -- three in-order waves per SIMD that each carry 17 issue slots per MFMA block
// each other.  Only 256 of the 410 VALU instructions are the Softplus; the rest was addressing, bounds selects, accumulator copies
// and 54 single-word operand reads per tile.  Here:
//   * rows are loaded through wave-uniform base pointers + ONE 32-bit lane offset (global_load ... saddr), indices CLAMPED to the
//     last point instead of selected (rows past n compute garbage that is never stored) -- no 64-bit lane arithmetic, no selects;
//   * the tile sits in LDS operand-major, [k & 1][row][k >> 1] with 20-float rows, so a lane fetches the A operands of four
//     k-steps with one ds_read_b128 (5 reads instead of 18), and the weights the same way (10 instead of 36);
//   * the loop is unrolled over the two accumulator sets (no copy), no exec-mask branches in the steady state;
//   * the reduce scratch has a 36-float row stride: 4 x ds_read_b128 instead of 16 single reads.
//   * the Softplus costs 4 instead of 6 full-rate instructions per activation: W1 and b1 are staged in LDS pre-multiplied by
//     100 log2(e) and the bias rides in the pad column (input 1.0), so the accumulator IS y = 100 log2(e) (W1 x + b1), and
//       softplus_100(z) = ln2 / 100 * (max(y, 0) + log2(1 + 2^-|y|))      [v_exp_f32 takes -|y| as source modifiers]
//     with ln2 / 100 folded into the output weight.  fp32 MFMA and the fp32 VALU issue against the same 64 FLOP/clk/SIMD
//     (counters: matrix-pipe busy + VALU busy = 94 .. 97 % of the launch in every variant of this kernel), so every VALU
//     instruction removed is time removed.  Against the kernel above: the products round differently (1e-7 relative).
constexpr int S2_LDK = 20;                 // operand row: 18 k-steps + 2 pad floats (80 B: 16-byte aligned, conflict-free b128)
constexpr int S2_BUF = 64 * S2_LDK;        // floats per tile buffer: operand rows (k & 1) * 32 + row
constexpr int S2_LDR = 36;                 // reduce scratch (inside a consumed tile buffer): 32 rows x 32 partials
constexpr float S2_C = 144.26950408889634f;        // 100 log2(e)

__device__ __forceinline__ float softplus_y_times(float y, float w, float part)      // w = ln2 / 100 * output weight
{
#ifdef IA_SOFTPLUS_SKIP
    // experiment (round 5): skip the two transcendentals when EVERY lane of the wave is saturated (|y| >= 24: 2^-|y| is
    // below half an ulp of 1, log2(1 + e) = 0 exactly in fp32).  Measured with tools/sdf_head_probe.py -- see DESIGN 4.4.
    float l = 0.0f;
    if (__ballot(fabsf(y) < 24.0f) != 0ull) {
        const float e_ = __builtin_amdgcn_exp2f(-fabsf(y));
        l = __builtin_amdgcn_logf(1.0f + e_);
    }
    float m_;
    asm("v_max_f32 %0, 0, %1" : "=v"(m_) : "v"(y));
    return fmaf(m_ + l, w, part);
#else
    const float e = __builtin_amdgcn_exp2f(-fabsf(y));
    const float l = __builtin_amdgcn_logf(1.0f + e);
    float m;                                                     // max(y, 0): fmaxf() would canonicalise the MFMA result first (2 instructions)
    asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(y));
    return fmaf(m + l, w, part);
#endif
}

struct S2Rows { float2 q[8]; float v[2]; };

template <int SH_WAVES>
__global__ __launch_bounds__(SH_WAVES * 64) void sdf_head_pipelined2_kernel(int64_t n, const float2* __restrict__ levels,
                                                                            const float* __restrict__ xp, const float* __restrict__ W1,
                                                                            const float* __restrict__ b1, const float* __restrict__ Wo,
                                                                            const float* __restrict__ bo, float* __restrict__ sdf)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;                                            // [2][64][S2_LDK]: sW[((k & 1) * 64 + h) * S2_LDK + (k >> 1)] = W1[h][k]
    float* sXall = sW + 2 * 64 * S2_LDK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const sX0 = sXall + wave * S2_BUF;                    // ONE tile buffer per wave: rows of the next tile -> reduce scratch -> rows of the one after
    for (int i = tid; i < 2 * 64 * S2_LDK; i += SH_WAVES * 64) {
        const int par = i / (64 * S2_LDK), rem = i - par * (64 * S2_LDK), h = rem / S2_LDK, kk = rem - h * S2_LDK, k = 2 * kk + par;
        sW[i] = (kk < 18 && k < 35) ? S2_C * W1[h * 35 + k] : (kk == 17 && k == 35) ? S2_C * b1[h] : 0.0f;
    }
    for (int i = lane; i < S2_BUF; i += 64) sX0[i] = 0.0f;
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    const float w0 = 0.0069314718055994531f * Wo[lr], w1 = 0.0069314718055994531f * Wo[32 + lr], bout = bo[0];
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * SH_WAVES;
    int64_t tile = (int64_t)blockIdx.x * SH_WAVES + wave;        // wave-uniform
    if (tile >= n_tiles) return;

    // loop-invariant lane parts of the addresses
    const uint32_t lvl_half = (uint32_t)lk * (uint32_t)(n * 8);               // the odd level of a load's level pair (n < 2^28)
    const uint32_t xi0 = (uint32_t)lane, xi1 = min(64u + (uint32_t)lane, 95u);    // xyz scalars 0 .. 95 of the tile
    const int st_row = lr * S2_LDK;                                             // feature (row, level l): [st_row + l] and [+ 32 rows]
    const int sx0 = ((xi0 % 3u) & 1u) * 32 * S2_LDK + (xi0 / 3u) * S2_LDK + 16 + ((xi0 % 3u) >> 1);
    const int sx1 = ((xi1 % 3u) & 1u) * 32 * S2_LDK + (xi1 / 3u) * S2_LDK + 16 + ((xi1 % 3u) >> 1);
    const int a_off = (lk * 32 + lr) * S2_LDK;
    const float* const wp0 = sW + (lk * 64 + lr) * S2_LDK;
    const float* const wp1 = wp0 + 32 * S2_LDK;
    const char* const lvb = reinterpret_cast<const char*>(levels);
    const char* const xpb = reinterpret_cast<const char*>(xp);

    auto load_rows = [&](S2Rows& r, int64_t t) {
        const int64_t tc = t < n_tiles ? t : n_tiles - 1;                      // past the end: the last tile again (never stored)
        const int64_t p0 = tc * 32;
        const uint32_t rem = (uint32_t)min((int64_t)32, n - p0);               // >= 1
        const uint32_t voff = lvl_half + ((uint32_t)p0 + min((uint32_t)lr, rem - 1u)) * 8u;
#pragma unroll
        for (int j = 0; j < 8; j++) r.q[j] = *reinterpret_cast<const float2*>(lvb + (size_t)(2 * j) * (size_t)n * 8 + voff);
        const uint32_t xb = (uint32_t)p0 * 3u, xm = rem * 3u - 1u;
        r.v[0] = *reinterpret_cast<const float*>(xpb + (size_t)((xb + min(xi0, xm)) * 4u));
        r.v[1] = *reinterpret_cast<const float*>(xpb + (size_t)((xb + min(xi1, xm)) * 4u));
    };
    auto store_rows = [&](const S2Rows& r, float* sT) {          // (no __restrict__ on the LDS pointers: the three uses alias on purpose)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int lvl = 2 * j + lk;
            sT[st_row + lvl] = r.q[j].x; sT[32 * S2_LDK + st_row + lvl] = r.q[j].y;
        }
        sT[sx0] = r.v[0] * 2.0f - 1.0f;
        sT[sx1] = r.v[1] * 2.0f - 1.0f;                                        // lanes 32 .. 63 repeat scalar 95
        sT[32 * S2_LDK + st_row + 17] = 1.0f;                                  // column 35: the bias column
    };
    // layer 1 of the tile in sN into (d0, d1); with SOFT the Softplus + output-layer partials of (s0, s1) run under the MFMAs
    auto layer1 = [&](const float* sN, f32x16& d0, f32x16& d1, const f32x16& s0, const f32x16& s1, float* part, auto soft) {
        constexpr bool SOFT = decltype(soft)::value;
        const float* ap = sN + a_off;
        f32x4 a4 = *reinterpret_cast<const f32x4*>(ap), b4 = *reinterpret_cast<const f32x4*>(wp0), c4 = *reinterpret_cast<const f32x4*>(wp1);
#pragma unroll
        for (int r = 0; r < 16; r++) { d0[r] = 0.f; d1[r] = 0.f; }
#pragma unroll
        for (int g = 0; g < 5; g++) {
            f32x4 an = a4, bn = b4, cn = c4;
            if (g < 3) {
                an = *reinterpret_cast<const f32x4*>(ap + 4 * (g + 1)); bn = *reinterpret_cast<const f32x4*>(wp0 + 4 * (g + 1));
                cn = *reinterpret_cast<const f32x4*>(wp1 + 4 * (g + 1));
            } else if (g == 3) {
                const float2 a2 = *reinterpret_cast<const float2*>(ap + 16), b2 = *reinterpret_cast<const float2*>(wp0 + 16),
                             c2 = *reinterpret_cast<const float2*>(wp1 + 16);
                an[0] = a2.x; an[1] = a2.y; bn[0] = b2.x; bn[1] = b2.y; cn[0] = c2.x; cn[1] = c2.y;
            }
#pragma unroll
            for (int q = 0; q < (g < 4 ? 4 : 2); q++) {
                const int kk = 4 * g + q;
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], b4[q], d0, 0, 0, 0);
                if (SOFT && kk < 16) part[kk] = softplus_y_times(s0[kk], w0, 0.0f);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], c4[q], d1, 0, 0, 0);
                if (SOFT && kk < 16) part[kk] = softplus_y_times(s1[kk], w1, part[kk]);
                if (SOFT) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);       // one activation of VALU work
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
            }
            a4 = an; b4 = bn; c4 = cn;
        }
    };
    using True = std::integral_constant<bool, true>;
    using False = std::integral_constant<bool, false>;
    // output layer of the tile whose partials are in part[]: transpose-reduce through the consumed tile buffer sR, store
    auto finish = [&](const float* part, float* sR, int64_t t) {
#pragma unroll
        for (int r = 0; r < 16; r++) sR[((r & 3) + 8 * (r >> 2) + 4 * lk) * S2_LDR + lr] = part[r];
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sR + lr * S2_LDR + lk * 16 + 4 * q);
            sum += v[0]; sum += v[1]; sum += v[2]; sum += v[3];
        }
        sum += __shfl_xor(sum, 32, 64);
        const int64_t p = t * 32 + lr;
        if (lk == 0 && p < n) sdf[p] = sum + bout;
    };

    f32x16 accA0, accA1, accB0, accB1;
    S2Rows rows;
    float part[16];
    // the wave's LDS buffer in program order (LDS operations of one wave execute in order): rows of tile + stride (read by layer 1)
    // -> scratch of the output-layer reduce of tile -> rows of tile + 2 stride (from the registers loaded at the top of the phase)
    load_rows(rows, tile);
    store_rows(rows, sX0);
    load_rows(rows, tile + stride);
    layer1(sX0, accA0, accA1, accA0, accA1, part, False{});
    store_rows(rows, sX0);
    for (;;) {
        load_rows(rows, tile + 2 * stride);
        layer1(sX0, accB0, accB1, accA0, accA1, part, True{});          // Softplus(accA) of tile under layer 1 of tile + stride -> accB
        finish(part, sX0, tile);
        tile += stride;
        if (tile >= n_tiles) break;
        store_rows(rows, sX0);
        load_rows(rows, tile + 2 * stride);
        layer1(sX0, accA0, accA1, accB0, accB1, part, True{});
        finish(part, sX0, tile);
        tile += stride;
        if (tile >= n_tiles) break;
        store_rows(rows, sX0);
    }
}

// (tried: the same pipeline with the level-major features DMA'd straight into LDS -- gfx950 global_load_lds_dwordx4, the tile kept
//  level-major so that the layer-1 A operand needs no transposition, nothing of the next-next tile in registers: 3.25 ms per
//  50 M points at 12 waves per CU, exactly what this kernel does at 12 waves once both reached it without spills -- the row
//  loads were not what limits it.  Not kept.)
template <int SH_WAVES, bool PROBE_CACHED = false>
static int launch_sdf_head_pipelined(int64_t n, const void* levels, const float* xp, const float* W1, const float* b1, const float* Wo,
                                     const float* bo, float* sdf, hipStream_t s)
{
    constexpr size_t lds = sizeof(float) * (64 * SH_LDW + SH_WAVES * 2 * 32 * SH_LDX);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)sdf_head_pipelined_kernel<SH_WAVES, PROBE_CACHED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int64_t n_tiles = (n + 31) / 32;
    int grid = (int)((n_tiles + SH_WAVES - 1) / SH_WAVES);
    if (grid > 256) grid = 256;
    sdf_head_pipelined_kernel<SH_WAVES, PROBE_CACHED><<<grid, SH_WAVES * 64, lds, s>>>(n, reinterpret_cast<const float2*>(levels), xp, W1, b1, Wo, bo, sdf);
    return ia::check_launch("ia_sdf_levels_fwd");
}

template <int SH_WAVES>
static int launch_sdf_head_pipelined2(int64_t n, const void* levels, const float* xp, const float* W1, const float* b1, const float* Wo,
                                      const float* bo, float* sdf, hipStream_t s)
{
    constexpr size_t lds = sizeof(float) * (2 * 64 * S2_LDK + SH_WAVES * S2_BUF);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(32 * S2_LDR <= S2_BUF, "the reduce scratch lives in a tile buffer");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)sdf_head_pipelined2_kernel<SH_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int64_t n_tiles = (n + 31) / 32;
    int grid = (int)((n_tiles + SH_WAVES - 1) / SH_WAVES);
    if (grid > 256) grid = 256;
    sdf_head_pipelined2_kernel<SH_WAVES><<<grid, SH_WAVES * 64, lds, s>>>(n, reinterpret_cast<const float2*>(levels), xp, W1, b1, Wo, bo, sdf);
    return ia::check_launch("ia_sdf_levels_fwd");
}

}  // namespace

// kind: 0 = SDF 35->64->13 softplus100 (optionally with analytic gradient)
//       1 = radiance 67->64->64->3 relu, sigmoid output
//       2 = material 48->64->64->5 relu, sigmoid output
IA_EXPORT int ia_mlp_fwd(int kind, int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
                         const int* seg_width, const float* seg_mul, const float* seg_add, const float* W1,
                         const float* b1, const float* W2, const float* b2, const float* Wo, const float* bo,
                         float* y, int y_stride, const float* jac, int xyz_col, const float* inv_scale_host,
                         float* grad, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    MlpArgs a = {};
    const int in_dim = kind == 0 ? 35 : (kind == 1 ? 67 : 48);
    IA_REQUIRE(kind >= 0 && kind <= 2, "unknown MLP kind");
    a.n = n;
    a.n_segs = n_segs;
    int r = mlp::fill_segs(a.segs, kind, n_segs, seg_ptr, seg_stride, seg_width, seg_mul, seg_add);
    if (r != IA_OK) return r;
    (void)in_dim;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Wo = Wo; a.bo = bo;
    a.y = y; a.y_stride = y_stride;
    a.jac = jac; a.xyz_col = xyz_col; a.grad = grad;
    if (inv_scale_host) for (int k = 0; k < 3; k++) a.inv_scale[k] = inv_scale_host[k];
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) {
        if (grad) {
            IA_REQUIRE(jac != nullptr && inv_scale_host != nullptr, "SDF gradient needs the hash-grid Jacobian and 1/scale");
            IA_REQUIRE(seg_width[0] == 32, "SDF head: segment 0 must be the 32 hash features");
            return launch_fwd<0, 35, 1, 13, 1, 0, true>(a, s);
        }
        return launch_fwd<0, 35, 1, 13, 1, 0, false>(a, s);
    }
    if (kind == 1) return launch_fwd<1, 67, 2, 3, 0, 1, false>(a, s);
    return launch_fwd<2, 48, 2, 5, 0, 1, false>(a, s);
}

// SDF value only, from the level-major hash features (no-grad coarse queries: importance resampling passes, the secondary
// march): levels = float2 [16][n] as left in the scratch of ia_hashgrid_fwd_xcd(out = NULL), xp [n,3] in [0,1].
// Same first layer / Softplus as kind 0; only row 0 of the output layer (the SDF, rf/geometry.py:152-160) is evaluated
// and 4 bytes per point are written instead of the 13-wide feature row.
IA_EXPORT int ia_sdf_levels_fwd(int64_t n, const void* levels, const float* xp, const float* W1, const float* b1, const float* Wo,
                                const float* bo, float* sdf, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n < ((int64_t)1 << 31), "ia_sdf_levels_fwd: at most 2^31 points per call");
    {
        // default ("pipe2"): two tiles per wave, 16 waves per CU (117 VGPRs, 5 KB of LDS per wave), operand-major LDS tiles, rescaled
        // Softplus: 2.81 ms per 50 M points = 0.52 of the fp32 MFMA peak (tools/sdf_head_ab.py; 12 waves 2.87, 8 waves 2.97); its
        // predecessor "pipe12" (162 VGPRs, 12 waves): 3.25 ms (0.45); the one-tile-per-wave kernel "tile": 4.80 ms (0.31); "pipe8":
        // 3.45 ms.  pipe2 addresses a level pair with a 32-bit lane offset: batches of 2^28 points and more take pipe12.
        const char* e = getenv("IA_SDF_HEAD");
        if (e && !strcmp(e, "probe_cached")) return launch_sdf_head_pipelined<12, true>(n, levels, xp, W1, b1, Wo, bo, sdf, (hipStream_t)stream);
        if (e && !strcmp(e, "pipe8")) return launch_sdf_head_pipelined<8>(n, levels, xp, W1, b1, Wo, bo, sdf, (hipStream_t)stream);
        if ((e && !strcmp(e, "pipe12")) || (!(e && e[0] == 't') && n >= ((int64_t)1 << 28)))
            return launch_sdf_head_pipelined<12>(n, levels, xp, W1, b1, Wo, bo, sdf, (hipStream_t)stream);
        if (e && !strcmp(e, "pipe2w12")) return launch_sdf_head_pipelined2<12>(n, levels, xp, W1, b1, Wo, bo, sdf, (hipStream_t)stream);
        if (e && !strcmp(e, "pipe2w8")) return launch_sdf_head_pipelined2<8>(n, levels, xp, W1, b1, Wo, bo, sdf, (hipStream_t)stream);
        if (!(e && e[0] == 't')) return launch_sdf_head_pipelined2<16>(n, levels, xp, W1, b1, Wo, bo, sdf, (hipStream_t)stream);
    }
    MlpArgs a = {};
    a.n = n;
    a.n_segs = 2;
    for (int s = 0; s < MAX_SEGS; s++) { a.segs[s].p = nullptr; a.segs[s].stride = 0; a.segs[s].width = 0; a.segs[s].mul = 1.f; a.segs[s].add = 0.f; }
    a.segs[0].p = (const float*)levels; a.segs[0].stride = (int)n; a.segs[0].width = 32;
    a.segs[1].p = xp; a.segs[1].stride = 3; a.segs[1].width = 3; a.segs[1].mul = 2.0f; a.segs[1].add = -1.0f;
    a.W1 = W1; a.b1 = b1; a.W2 = nullptr; a.b2 = nullptr; a.Wo = Wo; a.bo = bo;
    a.y = sdf; a.y_stride = 1;
    return launch_fwd<3, 35, 1, 1, 1, 0, false>(a, (hipStream_t)stream);
}

// The full SDF head (13 outputs + analytic gradient, kind 0 of ia_mlp_fwd with grad) on the level-major results of the
// XCD-partitioned gather: levels = float2 [16][n], levels_jac = float [16][n][6] (d feature / d x', features 2l and 2l+1),
// both as ia_hashgrid_fwd_levels leaves them in its scratch.  No [n,32] rows, no [n,32,3] Jacobian.
IA_EXPORT int ia_sdf_levels_fwd_grad(int64_t n, const void* levels, const float* levels_jac, const float* xp, const float* W1,
                                     const float* b1, const float* Wo, const float* bo, float* y, int y_stride,
                                     const float* inv_scale_host, float* grad, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n < ((int64_t)1 << 31), "ia_sdf_levels_fwd_grad: at most 2^31 points per call");
    IA_REQUIRE(levels_jac != nullptr && inv_scale_host != nullptr && grad != nullptr, "ia_sdf_levels_fwd_grad: Jacobian, 1/scale and grad are required");
    IA_REQUIRE(y_stride >= 13, "ia_sdf_levels_fwd_grad: y_stride must be >= 13");
    MlpArgs a = {};
    a.n = n;
    a.n_segs = 2;
    for (int s = 0; s < MAX_SEGS; s++) { a.segs[s].p = nullptr; a.segs[s].stride = 0; a.segs[s].width = 0; a.segs[s].mul = 1.f; a.segs[s].add = 0.f; }
    a.segs[0].p = (const float*)levels; a.segs[0].stride = (int)n; a.segs[0].width = 32;
    a.segs[1].p = xp; a.segs[1].stride = 3; a.segs[1].width = 3; a.segs[1].mul = 2.0f; a.segs[1].add = -1.0f;
    a.W1 = W1; a.b1 = b1; a.W2 = nullptr; a.b2 = nullptr; a.Wo = Wo; a.bo = bo;
    a.y = y; a.y_stride = y_stride;
    a.jac = levels_jac; a.xyz_col = 32; a.grad = grad;
    for (int k = 0; k < 3; k++) a.inv_scale[k] = inv_scale_host[k];
    return launch_fwd<3, 35, 1, 13, 1, 0, true>(a, (hipStream_t)stream);
}
