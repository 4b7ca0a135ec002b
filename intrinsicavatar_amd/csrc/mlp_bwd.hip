// mlp_bwd.hip -- backward (data path) of the fused MLPs on the CDNA4 matrix cores (fp32 MFMA).
//
// One kernel per MLP kind recomputes the forward tile-by-tile (64 points per wave, ONE in-place LDS
// tile per wave) and back-propagates through the layers with the same MFMA tile code as mlp.hip:
//     forward orientation   C = T . W^T   (a = T[p][k],  b = W[j][k])
//     backward orientation  C = T . W     (a = T[p][k],  b = W[k][j])
// It emits   * g_x  : gradient w.r.t. the assembled input row (hash features, geometry feature, SH,
//                     normal ...), consumed by ia_hashgrid_bwd / the upstream kernels;
//            * the per-layer operand pairs (layer input, pre-activation gradient) in HBM, from which
//              the tiny weight gradients dW_l = G_l^T A_{l-1} are formed by a plain library GEMM
//              (rocBLAS via torch.matmul on the host side) -- a [64 x n] x [n x 68] reduction.
// SDF head (kind 0) additionally carries the SECOND-ORDER terms needed because the analytic normal
// d sdf / d x is an output that losses depend on (eikonal, normal-conditioned radiance): the
// reference gets them from autograd's double backward through VanillaMLP + tcnn
// (models/rf/geometry.py:165-172 with create_graph=True).
#include "mlp_tile.h"

namespace {

using mlp::Seg;
using mlp::MAX_SEGS;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int HID = 64;

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2])
{
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][nt][r] = 0.0f;
}

// acc[m][nt] += T[64 x 2*KSTEPS] . W^T   with W row-major [64][ldw]
template <int KSTEPS>
__device__ __forceinline__ void gemm_xwT(const float* sT, int ldx, const float* sW, int ldw, f32x16 (&acc)[2][2], int lane)
{
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll 2
    for (int kk = 0; kk < KSTEPS; kk++) {
        const int k = 2 * kk + lk;
        const float a0 = sT[lr * ldx + k], a1 = sT[(32 + lr) * ldx + k];
        const float b0 = sW[lr * ldw + k], b1 = sW[(32 + lr) * ldw + k];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// acc[m][nt] += T[64 x 2*KSTEPS] . W[:, col0 : col0+64]   with W row-major [K][ldw]; columns >= ncols read as 0
template <int KSTEPS>
__device__ __forceinline__ void gemm_xw(const float* sT, int ldx, const float* sW, int ldw, int col0, int ncols,
                                        f32x16 (&acc)[2][2], int lane)
{
    const int lr = lane & 31, lk = lane >> 5;
    const int c0 = col0 + lr, c1 = col0 + 32 + lr;
#pragma unroll 2
    for (int kk = 0; kk < KSTEPS; kk++) {
        const int k = 2 * kk + lk;
        const float a0 = sT[lr * ldx + k], a1 = sT[(32 + lr) * ldx + k];
        const float b0 = c0 < ncols ? sW[k * ldw + c0] : 0.0f;
        const float b1 = c1 < ncols ? sW[k * ldw + c1] : 0.0f;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
}

#define ACC_FOREACH(m, nt, r, row, col, lane)                                   \
    _Pragma("unroll") for (int m = 0; m < 2; m++)                               \
    _Pragma("unroll") for (int nt = 0; nt < 2; nt++)                            \
    _Pragma("unroll") for (int r = 0; r < 16; r++)                              \
        if (const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * ((lane) >> 5); true) \
            if (const int col = 32 * nt + ((lane) & 31); true)

__device__ __forceinline__ void stage_matrix(float* dst, int ld, const float* src, int rows, int cols, int rows_pad,
                                             int tid)
{
    for (int i = tid; i < rows_pad * ld; i += THREADS) {
        const int r = i / ld, c = i % ld;
        dst[i] = (r < rows && c < cols) ? src[r * cols + c] : 0.0f;
    }
}

// copy the [64 x cols] LDS tile to global [n, gstride] (rows beyond n skipped); coalesced along columns
__device__ __forceinline__ void tile_to_global(const float* sT, int ldx, float* g, int gstride, int cols, int64_t p0,
                                               int64_t n, int lane)
{
    if (!g) return;
    const int tot = 64 * cols;
    for (int i = lane; i < tot; i += 64) {
        const int r = i / cols, c = i % cols;
        const int64_t p = p0 + r;
        if (p < n) g[p * gstride + c] = sT[r * ldx + c];
    }
}

// ------------------------------------------------------------------------------------------------
// ReLU MLPs with two hidden layers and a sigmoid output (radiance 67->64->64->3, material 48->64->64->5)
struct Bwd2Args {
    int64_t n;
    int n_segs;
    Seg segs[MAX_SEGS];
    const float *W1, *b1, *W2, *b2, *Wo, *bo;
    const float* g_y;     // [n, OUT]  gradient w.r.t. the (sigmoid) output
    float* g_x;           // [n, gx_stride]  gradient w.r.t. the assembled input row (IN columns)
    int gx_stride;
    // operand pairs for the weight-gradient GEMMs (all optional)
    float *X, *A1, *A2;   // [n, IN_PAD], [n,64], [n,64]
    float *G1, *G2, *G3;  // [n,64], [n,64], [n,16]
};

template <int KIND, int IN, int OUT>
__global__ __launch_bounds__(THREADS) void mlp2_bwd_kernel(Bwd2Args a)
{
    constexpr int IN_PAD = (IN + 1) / 2 * 2;
    constexpr int LDW1 = IN_PAD + 1, LDW = HID + 1, LDX = (IN_PAD > HID ? IN_PAD : HID) + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;
    float* sW2 = sW1 + HID * LDW1;
    float* sWo = sW2 + HID * LDW;
    float* sB = sWo + 16 * LDW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sT = sB + 144 + wave * 64 * LDX;
    stage_matrix(sW1, LDW1, a.W1, HID, IN, HID, tid);
    stage_matrix(sW2, LDW, a.W2, HID, HID, HID, tid);
    stage_matrix(sWo, LDW, a.Wo, OUT, HID, 16, tid);
    if (tid < 64) { sB[tid] = a.b1[tid]; sB[64 + tid] = a.b2[tid]; }
    if (tid < 16) sB[128 + tid] = (tid < OUT) ? a.bo[tid] : 0.0f;
    __syncthreads();

    const int64_t n_tiles = (a.n + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t p0 = tile * 64;
        mlp::assemble<KIND, IN>(sT, LDX, a.segs, p0, a.n, lane);
        tile_to_global(sT, LDX, a.X, IN_PAD, IN_PAD, p0, a.n, lane);
        f32x16 acc[2][2];
        unsigned long long m1 = 0ull, m2 = 0ull;      // relu masks in accumulator-fragment order
        // ---- forward ----
        zero_acc(acc);
        gemm_xwT<IN_PAD / 2>(sT, LDX, sW1, LDW1, acc, lane);
        {
            int bit = 0;
            ACC_FOREACH(m, nt, r, row, col, lane) {
                const float v = fmaxf(acc[m][nt][r] + sB[col], 0.0f);
                sT[row * LDX + col] = v;
                if (v > 0.0f) m1 |= 1ull << bit;
                bit++;
            }
        }
        tile_to_global(sT, LDX, a.A1, HID, HID, p0, a.n, lane);
        zero_acc(acc);
        gemm_xwT<HID / 2>(sT, LDX, sW2, LDW, acc, lane);
        {
            int bit = 0;
            ACC_FOREACH(m, nt, r, row, col, lane) {
                const float v = fmaxf(acc[m][nt][r] + sB[64 + col], 0.0f);
                sT[row * LDX + col] = v;
                if (v > 0.0f) m2 |= 1ull << bit;
                bit++;
            }
        }
        tile_to_global(sT, LDX, a.A2, HID, HID, p0, a.n, lane);
        // output layer + G3 = g_y * y (1 - y)
        {
            const int l15 = lane & 15, l4 = lane >> 4;
            f32x4 o[4];
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) o[m][r] = 0.0f;
#pragma unroll 4
            for (int kk = 0; kk < HID / 4; kk++) {
                const int k = 4 * kk + l4;
                const float b = sWo[l15 * LDW + k];
#pragma unroll
                for (int m = 0; m < 4; m++)
                    o[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(sT[(16 * m + l15) * LDX + k], b, o[m], 0, 0, 0);
            }
            // all reads of A2 are done (in-order wave): overwrite columns 0..15 of the tile with G3
            const float bias = sB[128 + l15];
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * m + l4 * 4 + r;
                    const int64_t p = p0 + row;
                    float g = 0.0f;
                    if (l15 < OUT && p < a.n) {
                        const float y = 1.0f / (1.0f + __expf(-(o[m][r] + bias)));
                        g = a.g_y[p * OUT + l15] * y * (1.0f - y);
                    }
                    sT[row * LDX + l15] = g;
                }
        }
        tile_to_global(sT, LDX, a.G3, 16, 16, p0, a.n, lane);
        // ---- backward ----
        zero_acc(acc);
        gemm_xw<8>(sT, LDX, sWo, LDW, 0, HID, acc, lane);          // GA2 = G3[64x16] . Wo[16x64]
        {
            int bit = 0;
            ACC_FOREACH(m, nt, r, row, col, lane) {
                sT[row * LDX + col] = ((m2 >> bit) & 1ull) ? acc[m][nt][r] : 0.0f;
                bit++;
            }
        }
        tile_to_global(sT, LDX, a.G2, HID, HID, p0, a.n, lane);
        zero_acc(acc);
        gemm_xw<HID / 2>(sT, LDX, sW2, LDW, 0, HID, acc, lane);    // GA1 = G2 . W2
        {
            int bit = 0;
            ACC_FOREACH(m, nt, r, row, col, lane) {
                sT[row * LDX + col] = ((m1 >> bit) & 1ull) ? acc[m][nt][r] : 0.0f;
                bit++;
            }
        }
        tile_to_global(sT, LDX, a.G1, HID, HID, p0, a.n, lane);
        // g_x = G1 . W1   [64 x IN], 64 columns at a time straight to global
        if (a.g_x) {
#pragma unroll
            for (int c0 = 0; c0 < IN; c0 += 64) {
                zero_acc(acc);
                gemm_xw<HID / 2>(sT, LDX, sW1, LDW1, c0, IN, acc, lane);
                ACC_FOREACH(m, nt, r, row, col, lane) {
                    const int64_t p = p0 + row;
                    if (p < a.n && c0 + col < IN) a.g_x[p * a.gx_stride + c0 + col] = acc[m][nt][r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SDF head 35->64->13 (Softplus beta=100) with first- AND second-order terms.
//   forward   z = W1 h + b1, a = sp(z), s = sigmoid(100 z), out = W2 a + b2, gz = s * W2[0,:], gh = gz W1
//   given     g_out [n,13]  (d L / d out; the SDF gradient is g_out[:,0])
//             q     [n,3]   (d L / d (d sdf / d x'))  = g_grad * inv_scale
//   u  = [J q (32) | 2 q (3)]                  (JVP of the input row along q)
//   dgz = u W1^T ;  da = g_out W2 ;  dz = da * s + dgz * W2[0,:] * 100 s (1 - s)
//   outputs  gE = (dz W1)[:, :32]  (d L / d hash features, first order)
//            gG = gh[:, :32]       (coefficients of the second-order table scatter, paired with q)
//            operand pairs for the weight GEMMs:  dW1 = dz^T h + gz^T u ; dW2 = g_out^T a (+ row0 += sum dgz*s)
struct SdfBwdArgs {
    int64_t n;
    int n_segs;
    Seg segs[MAX_SEGS];
    const float *W1, *b1, *Wo, *bo;
    const float* jac;      // [n,32,3]
    const float* g_out;    // [n,13]
    const float* q;        // [n,3]
    float *gE, *gG;        // [n,32] each
    float *Hh, *U, *DZ, *GZ, *A, *DGS;   // [n,36] [n,36] [n,64] [n,64] [n,64] [n,64]
};

__global__ __launch_bounds__(THREADS) void sdf_bwd_kernel(SdfBwdArgs a)
{
    constexpr int IN = 35, IN_PAD = 36, OUT = 13;
    constexpr int LDW1 = IN_PAD + 1, LDW = HID + 1, LDX = HID + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;
    float* sWo = sW1 + HID * LDW1;
    float* sB = sWo + 16 * LDW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sT = sB + 144 + wave * 64 * LDX;
    stage_matrix(sW1, LDW1, a.W1, HID, IN, HID, tid);
    stage_matrix(sWo, LDW, a.Wo, OUT, HID, 16, tid);
    if (tid < 64) sB[tid] = a.b1[tid];
    if (tid < 16) sB[128 + tid] = (tid < OUT) ? a.bo[tid] : 0.0f;
    __syncthreads();

    const int64_t n_tiles = (a.n + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t p0 = tile * 64;
        f32x16 acc[2][2];
        float sig[2][2][16];
        // ---- forward: h -> z -> (a, s) ----
        mlp::assemble<0, IN>(sT, LDX, a.segs, p0, a.n, lane);
        tile_to_global(sT, LDX, a.Hh, IN_PAD, IN_PAD, p0, a.n, lane);
        zero_acc(acc);
        gemm_xwT<IN_PAD / 2>(sT, LDX, sW1, LDW1, acc, lane);
        ACC_FOREACH(m, nt, r, row, col, lane) {
            const float z = acc[m][nt][r] + sB[col];
            float sg;
            sT[row * LDX + col] = mlp::softplus100(z, sg);
            sig[m][nt][r] = sg;
        }
        tile_to_global(sT, LDX, a.A, HID, HID, p0, a.n, lane);
        // ---- gz = s * W2[0,:]  -> tile, global ; gh = gz W1 -> gG ----
        ACC_FOREACH(m, nt, r, row, col, lane) sT[row * LDX + col] = sig[m][nt][r] * sWo[col];
        tile_to_global(sT, LDX, a.GZ, HID, HID, p0, a.n, lane);
        zero_acc(acc);
        gemm_xw<HID / 2>(sT, LDX, sW1, LDW1, 0, IN_PAD, acc, lane);
        ACC_FOREACH(m, nt, r, row, col, lane) {
            const int64_t p = p0 + row;
            if (col < 32 && p < a.n) a.gG[p * 32 + col] = acc[m][nt][r];
        }
        // ---- u = [J q | 2 q] assembled per point (lane = point) ----
        {
            const int64_t p = p0 + lane;
            float* urow = sT + lane * LDX;
            if (p < a.n) {
                const float q0 = a.q[p * 3 + 0], q1 = a.q[p * 3 + 1], q2 = a.q[p * 3 + 2];
                const float* J = a.jac + p * 96;
#pragma unroll 8
                for (int k = 0; k < 32; k++) urow[k] = J[k * 3 + 0] * q0 + J[k * 3 + 1] * q1 + J[k * 3 + 2] * q2;
                urow[32] = 2.0f * q0; urow[33] = 2.0f * q1; urow[34] = 2.0f * q2; urow[35] = 0.0f;
            } else {
#pragma unroll 4
                for (int k = 0; k < IN_PAD; k++) urow[k] = 0.0f;
            }
        }
        tile_to_global(sT, LDX, a.U, IN_PAD, IN_PAD, p0, a.n, lane);
        // ---- dgz = u W1^T ; keep dgz*W2[0]*100 s(1-s) in registers, emit dgz*s ----
        zero_acc(acc);
        gemm_xwT<IN_PAD / 2>(sT, LDX, sW1, LDW1, acc, lane);
        float dz2[2][2][16];
        ACC_FOREACH(m, nt, r, row, col, lane) {
            const float s = sig[m][nt][r], d = acc[m][nt][r];
            dz2[m][nt][r] = d * sWo[col] * 100.0f * s * (1.0f - s);
            sT[row * LDX + col] = d * s;
        }
        tile_to_global(sT, LDX, a.DGS, HID, HID, p0, a.n, lane);
        // ---- da = g_out W2 : load g_out into tile columns 0..15 ----
        for (int i = lane; i < 64 * 16; i += 64) {
            const int r = i >> 4, c = i & 15;
            const int64_t p = p0 + r;
            sT[r * LDX + c] = (c < OUT && p < a.n) ? a.g_out[p * OUT + c] : 0.0f;
        }
        zero_acc(acc);
        gemm_xw<8>(sT, LDX, sWo, LDW, 0, HID, acc, lane);
        ACC_FOREACH(m, nt, r, row, col, lane)
            sT[row * LDX + col] = acc[m][nt][r] * sig[m][nt][r] + dz2[m][nt][r];
        tile_to_global(sT, LDX, a.DZ, HID, HID, p0, a.n, lane);
        // ---- gE = (dz W1)[:, :32] ----
        zero_acc(acc);
        gemm_xw<HID / 2>(sT, LDX, sW1, LDW1, 0, IN_PAD, acc, lane);
        ACC_FOREACH(m, nt, r, row, col, lane) {
            const int64_t p = p0 + row;
            if (col < 32 && p < a.n) a.gE[p * 32 + col] = acc[m][nt][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// split-K weight gradient:  dW[M x N] += G[n x M]^T . A[n x N]   (M <= 64, N <= 96; K = n points)
// rocBLAS maps this tall-skinny product to ONE output tile (a single workgroup walking K = millions);
// here K is split over the whole chip: each wave owns 16-row slabs, keeps the 2 x 3 MFMA accumulator
// tiles in registers across its slabs and flushes them once with atomics.
// The slabs are copied global -> LDS verbatim (row-major, caller's strides) with 16-byte loads: both MFMA
// operand reads (a = G[k][i], b = A[k][j], k = point) walk consecutive columns across lanes, which is
// conflict-free for ANY row stride, and columns beyond M / N only feed accumulator entries that are
// discarded, so no padding or masking is needed.
constexpr int WG_ROWS = 16;
__global__ __launch_bounds__(THREADS) void wgrad_kernel(int64_t n, const float* __restrict__ G, int gs, int M,
                                                        const float* __restrict__ A, int as, int N,
                                                        float* __restrict__ dW, int ldw, float* __restrict__ db)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g_tile = WG_ROWS * gs, a_tile = WG_ROWS * as;           // floats, multiples of 4
    const int per_wave = g_tile + 64 + a_tile + 96;
    float* sG = smem + wave * per_wave;
    float* sA = sG + g_tile + 64;
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[2][3];
    float colsum = 0.0f;      // db: lane c accumulates column c of G over all slabs of this wave
#pragma unroll
    for (int m = 0; m < 2; m++) {
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][nt][r] = 0.0f;
    }
    // slack words (read by out-of-range columns of the last rows) must be finite
    for (int i = lane; i < 64; i += 64) sG[g_tile + i] = 0.0f;
    for (int i = lane; i < 96; i += 64) sA[a_tile + i] = 0.0f;
    const int64_t n_tiles = (n + WG_ROWS - 1) / WG_ROWS;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t p0 = tile * WG_ROWS;
        const int rows = (int)((n - p0) < WG_ROWS ? (n - p0) : WG_ROWS);
        {
            const float4* src = reinterpret_cast<const float4*>(G + p0 * gs);
            float4* dst = reinterpret_cast<float4*>(sG);
            const int nv = rows * gs / 4;
            for (int i = lane; i < g_tile / 4; i += 64) dst[i] = i < nv ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (rows < WG_ROWS) for (int i = nv * 4 + lane; i < g_tile && i < rows * gs; i += 64) sG[i] = G[p0 * gs + i];
        }
        {
            const float4* src = reinterpret_cast<const float4*>(A + p0 * as);
            float4* dst = reinterpret_cast<float4*>(sA);
            const int nv = rows * as / 4;
            for (int i = lane; i < a_tile / 4; i += 64) dst[i] = i < nv ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (rows < WG_ROWS) for (int i = nv * 4 + lane; i < a_tile && i < rows * as; i += 64) sA[i] = A[p0 * as + i];
        }
#pragma unroll
        for (int kk = 0; kk < WG_ROWS / 2; kk++) {
            const int k = 2 * kk + lk;
            const float a0 = sG[k * gs + lr], a1 = sG[k * gs + 32 + lr];
            const float b0 = sA[k * as + lr], b1 = sA[k * as + 32 + lr], b2 = sA[k * as + 64 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, acc[0][2], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, acc[1][2], 0, 0, 0);
        }
        if (db && lane < M) {
#pragma unroll
            for (int k = 0; k < WG_ROWS; k++) colsum += sG[k * gs + lane];
        }
    }
    if (db && lane < M && colsum != 0.0f) unsafeAtomicAdd(db + lane, colsum);
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lk, col = 32 * nt + lr;
                const float v = acc[m][nt][r];
                if (row < M && col < N && v != 0.0f) unsafeAtomicAdd(dW + row * ldw + col, v);
            }
}

template <int KIND, int IN, int OUT>
int launch_bwd2(const Bwd2Args& a, hipStream_t s)
{
    constexpr int IN_PAD = (IN + 1) / 2 * 2;
    constexpr int LDW1 = IN_PAD + 1, LDW = HID + 1, LDX = (IN_PAD > HID ? IN_PAD : HID) + 1;
    constexpr size_t lds = sizeof(float) * (HID * LDW1 + HID * LDW + 16 * LDW + 144 + 4 * 64 * LDX);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = mlp2_bwd_kernel<KIND, IN, OUT>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int64_t n_tiles = (a.n + 63) / 64;
    int grid = (int)((n_tiles + 3) / 4);
    if (grid > 256) grid = 256;
    kern<<<grid, THREADS, lds, s>>>(a);
    return ia::check_launch("ia_mlp_bwd");
}

}  // namespace

// kind 1 (radiance 67->64->64->3) / 2 (material 48->64->64->5); see include/ia_amd.h
IA_EXPORT int ia_mlp_bwd(int kind, int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
                         const int* seg_width, const float* seg_mul, const float* seg_add, const float* W1,
                         const float* b1, const float* W2, const float* b2, const float* Wo, const float* bo,
                         const float* g_y, float* g_x, int gx_stride, float* X, float* A1, float* A2, float* G1,
                         float* G2, float* G3, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(kind == 1 || kind == 2, "ia_mlp_bwd: kind must be 1 (radiance) or 2 (material)");
    Bwd2Args a = {};
    a.n = n; a.n_segs = n_segs;
    int r = mlp::fill_segs(a.segs, kind, n_segs, seg_ptr, seg_stride, seg_width, seg_mul, seg_add);
    if (r != IA_OK) return r;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Wo = Wo; a.bo = bo;
    a.g_y = g_y; a.g_x = g_x; a.gx_stride = gx_stride;
    a.X = X; a.A1 = A1; a.A2 = A2; a.G1 = G1; a.G2 = G2; a.G3 = G3;
    return kind == 1 ? launch_bwd2<1, 67, 3>(a, (hipStream_t)stream) : launch_bwd2<2, 48, 5>(a, (hipStream_t)stream);
}

IA_EXPORT int ia_sdf_mlp_bwd(int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
                             const int* seg_width, const float* seg_mul, const float* seg_add, const float* W1,
                             const float* b1, const float* Wo, const float* bo, const float* jac, const float* g_out,
                             const float* q, float* gE, float* gG, float* Hh, float* U, float* DZ, float* GZ, float* A,
                             float* DGS, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    SdfBwdArgs a = {};
    a.n = n; a.n_segs = n_segs;
    int r = mlp::fill_segs(a.segs, 0, n_segs, seg_ptr, seg_stride, seg_width, seg_mul, seg_add);
    if (r != IA_OK) return r;
    a.W1 = W1; a.b1 = b1; a.Wo = Wo; a.bo = bo; a.jac = jac; a.g_out = g_out; a.q = q;
    a.gE = gE; a.gG = gG; a.Hh = Hh; a.U = U; a.DZ = DZ; a.GZ = GZ; a.A = A; a.DGS = DGS;
    constexpr int LDW1 = 37, LDW = 65, LDX = 65;
    constexpr size_t lds = sizeof(float) * (HID * LDW1 + 16 * LDW + 144 + 4 * 64 * LDX);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)sdf_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int64_t n_tiles = (n + 63) / 64;
    int grid = (int)((n_tiles + 3) / 4);
    if (grid > 256) grid = 256;
    sdf_bwd_kernel<<<grid, THREADS, lds, (hipStream_t)stream>>>(a);
    return ia::check_launch("ia_sdf_mlp_bwd");
}

// dW[M, ldw] += G[:, :M]^T A[:, :N]   (accumulated into; caller zeroes).  g_stride / a_stride in floats; rows must be
// 16-byte aligned per 16-row slab (any stride that is a multiple of 1 float works: 16 * stride * 4 B is a multiple of 16).
IA_EXPORT int ia_wgrad(int64_t n, const float* G, int g_stride, int M, const float* A, int a_stride, int N, float* dW,
                       int ldw, float* db, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(M >= 1 && M <= 64 && N >= 1 && N <= 96, "ia_wgrad: M <= 64, N <= 96");
    IA_REQUIRE(g_stride >= M && a_stride >= N && g_stride <= 64 && a_stride <= 96, "ia_wgrad: strides must be in [M,64] / [N,96]");
    IA_REQUIRE(((uintptr_t)G % 16) == 0 && ((uintptr_t)A % 16) == 0, "ia_wgrad: operands must be 16-byte aligned");
    const size_t lds = sizeof(float) * 4 * (WG_ROWS * g_stride + 64 + WG_ROWS * a_stride + 96);
    const int64_t n_tiles = (n + WG_ROWS - 1) / WG_ROWS;
    int grid = (int)((n_tiles + 3) / 4);
    if (grid > 768) grid = 768;
    wgrad_kernel<<<grid, THREADS, lds, (hipStream_t)stream>>>(n, G, g_stride, M, A, a_stride, N, dW, ldw, db);
    return ia::check_launch("ia_wgrad");
}
