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
