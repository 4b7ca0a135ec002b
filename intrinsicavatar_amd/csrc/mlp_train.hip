// mlp_train.hip -- fused backward of the field MLPs: data gradients AND weight gradients in one pass, nothing but
// the input-row gradient leaves the CU.
//
// mlp_bwd.hip (kept as the operand-emitting reference path) writes every layer's (input, pre-activation gradient) pair
// to HBM -- 1.4 KB per point and MLP -- and a second kernel (ia_wgrad) reads them back for dW = G^T A.  Here each wave
// keeps the operand tiles of its 32 points in LDS (three activation buffers that are recycled in the order the chain
// rule consumes them) and accumulates the weight gradients in MFMA accumulator registers across ALL of its tiles:
//
//     radiance / material (IN -> 64 -> 64 -> OUT, ReLU, sigmoid)              buffers
//       X -> B0 ; A1 = relu(X W1^T + b1) -> B1 ; A2 = relu(A1 W2^T + b2) -> B2
//       G3 = g_y * y (1 - y) -> Bg            dWo += G3^T A2   (Bg, B2)
//       G2 = (G3 Wo) . [A2 > 0] -> B2         dW2 += G2^T A1   (B2, B1)
//       G1 = (G2 W2) . [A1 > 0] -> B1         dW1 += G1^T X    (B1, B0)
//       g_x = G1 W1 -> global
//     dW* live in 176 accumulator registers per lane (6 + 4 tiles of 32x32, 4 tiles of 16x16); one wave per SIMD
//     (4 waves / CU, 150 KB of LDS), which is enough here because a tile now carries ~30 k MFMA cycles of work for one
//     input fetch.  At the end the four waves of a workgroup reduce their accumulators through LDS (plain read-modify-writes, one wave at a time) and
//     the workgroup adds its partial sums to the global dW with one atomic per element.
//
// The SDF head follows the same scheme with its second-order terms (see mlp_bwd.hip for the derivation):
//       dW1 = DZ^T H + GZ^T U , db1 = sum DZ , dWo = g_out^T A (+ row 0 += sum dgz * s) , dbo = sum g_out.
#include "mlp_tile.h"

namespace {

using mlp::Seg;
using mlp::MAX_SEGS;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int WAVES = 4;
constexpr int HID = 64;
constexpr int TM = 32;

template <int N>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[N])
{
#pragma unroll
    for (int t = 0; t < N; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
}

// acc[nt] += T[32 x 2*KSTEPS] . W^T   with W row-major [64][ldw]
template <int KSTEPS>
__device__ __forceinline__ void gemm_xwT(const float* sT, int ldx, const float* sW, int ldw, f32x16 (&acc)[2], int lane)
{
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll 2
    for (int kk = 0; kk < KSTEPS; kk++) {
        const int k = 2 * kk + lk;
        const float a0 = sT[lr * ldx + k];
        const float b0 = sW[lr * ldw + k], b1 = sW[(32 + lr) * ldw + k];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
    }
}

// acc[nt] += T[32 x 2*KSTEPS] . W[:, col0 : col0+64]   with W row-major [K][ldw]; columns >= ncols read as 0
template <int KSTEPS>
__device__ __forceinline__ void gemm_xw(const float* sT, int ldx, const float* sW, int ldw, int col0, int ncols,
                                        f32x16 (&acc)[2], int lane)
{
    const int lr = lane & 31, lk = lane >> 5;
    const int c0 = col0 + lr, c1 = col0 + 32 + lr;
#pragma unroll 2
    for (int kk = 0; kk < KSTEPS; kk++) {
        const int k = 2 * kk + lk;
        const float a0 = sT[lr * ldx + k];
        const float b0 = c0 < ncols ? sW[k * ldw + c0] : 0.0f;
        const float b1 = c1 < ncols ? sW[k * ldw + c1] : 0.0f;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
    }
}

// accW[m][nt] += G^T A over the tile's 32 points: G [32 x 64] (ldg), A [32 x ncols] (lda); k = point index, both
// operand reads walk consecutive columns of one row across lanes (conflict-free for any row stride)
template <int NT>
__device__ __forceinline__ void wgrad_acc(const float* sG, int ldg, const float* sA, int lda, int ncols,
                                          f32x16 (&accW)[2][NT], int lane)
{
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll 2
    for (int kk = 0; kk < TM / 2; kk++) {
        const int k = 2 * kk + lk;
        const float a0 = sG[k * ldg + lr], a1 = sG[k * ldg + 32 + lr];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int c = 32 * nt + lr;
            const float b = c < ncols ? sA[k * lda + c] : 0.0f;
            accW[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, accW[0][nt], 0, 0, 0);
            accW[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, accW[1][nt], 0, 0, 0);
        }
    }
}

// accO[nt] += G3^T A over 32 points with 16x16x4 tiles: G3 [32 x 16] (ldg), A [32 x 64] (lda)
__device__ __forceinline__ void wgrad_out_acc(const float* sG, int ldg, const float* sA, int lda, f32x4 (&accO)[4], int lane)
{
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll 2
    for (int kk = 0; kk < TM / 4; kk++) {
        const int k = 4 * kk + l4;
        const float a = sG[k * ldg + l15];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
            accO[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sA[k * lda + 16 * nt + l15], accO[nt], 0, 0, 0);
    }
}

#define ACC2_FOREACH(nt, r, row, col, lane)                                    \
    _Pragma("unroll") for (int nt = 0; nt < 2; nt++)                            \
    _Pragma("unroll") for (int r = 0; r < 16; r++)                              \
        if (const int row = (r & 3) + 8 * (r >> 2) + 4 * ((lane) >> 5); true)  \
            if (const int col = 32 * nt + ((lane) & 31); true)

__device__ __forceinline__ void stage_matrix(float* dst, int ld, const float* src, int rows, int cols, int rows_pad,
                                             int tid)
{
    for (int i = tid; i < rows_pad * ld; i += THREADS) {
        const int r = i / ld, c = i % ld;
        dst[i] = (r < rows && c < cols) ? src[r * cols + c] : 0.0f;
    }
}

__device__ __forceinline__ float col_sum(const float* sT, int ld, int col)
{
    float s = 0.0f;
#pragma unroll 8
    for (int r = 0; r < TM; r++) s += sT[r * ld + col];
    return s;
}

// workgroup reduction of per-wave accumulator tiles through an LDS scratch [64][LDR], then one global atomic per
// element (rows < M, cols < N).  Called by all threads.  The waves take turns adding their fragments with plain
// read-modify-writes: float LDS atomics (ds_add_f32) retire ~30x slower than integer ones on gfx950
// (tools/probes/lds_atomic_probe.hip), and a wave's fragment never aliases itself.
constexpr int LDR = 97;
template <int NT>
__device__ __forceinline__ void flush_w(float* sRed, const f32x16 (&accW)[2][NT], float* __restrict__ dW, int M, int N,
                                        int ldw, int tid)
{
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5;
    __syncthreads();
    for (int w = 0; w < WAVES; w++) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lk, col = 32 * nt + lr;
                        float* d = sRed + row * LDR + col;
                        *d = (w == 0) ? accW[m][nt][r] : *d + accW[m][nt][r];
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < M * N; i += THREADS) {
        const int row = i / N, col = i % N;
        const float v = sRed[row * LDR + col];
        if (v != 0.0f) unsafeAtomicAdd(dW + row * ldw + col, v);
    }
}

__device__ __forceinline__ void flush_o(float* sRed, const f32x4 (&accO)[4], float* __restrict__ dWo, int OUT, int tid)
{
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    __syncthreads();
    for (int w = 0; w < WAVES; w++) {
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float* d = sRed + (l4 * 4 + r) * LDR + 16 * nt + l15;
                    *d = (w == 0) ? accO[nt][r] : *d + accO[nt][r];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < OUT * HID; i += THREADS) {
        const int row = i / HID, col = i % HID;
        const float v = sRed[row * LDR + col];
        if (v != 0.0f) unsafeAtomicAdd(dWo + row * HID + col, v);
    }
}

// ------------------------------------------------------------------------------------------------
struct Train2Args {
    int64_t n;
    int n_segs;
    Seg segs[MAX_SEGS];
    const float *W1, *b1, *W2, *b2, *Wo, *bo;
    const float* g_y;     // [n, OUT]
    float* g_x;           // [n, gx_stride]
    int gx_stride;
    float *dW1, *db1, *dW2, *db2, *dWo, *dbo;     // accumulated into (caller zeroes): [64,IN] [64] [64,64] [64] [OUT,64] [OUT]
};

template <int KIND, int IN, int OUT>
__global__ __launch_bounds__(THREADS) void mlp2_train_kernel(Train2Args a)
{
    constexpr int IN_PAD = (IN + 1) / 2 * 2;
    constexpr int LDW1 = IN_PAD + 1, LDW = HID + 1;
    constexpr int LD0 = IN_PAD + 1, LDH = HID + 1, LDG = 17;
    constexpr int NT1 = (IN + 31) / 32;
    constexpr int PER_WAVE = TM * (LD0 + 2 * LDH + LDG);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;
    float* sW2 = sW1 + HID * LDW1;
    float* sWo = sW2 + HID * LDW;
    float* sB = sWo + 16 * LDW;
    float* sTiles = sB + 144;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* B0 = sTiles + wave * PER_WAVE;
    float* B1 = B0 + TM * LD0;
    float* B2 = B1 + TM * LDH;
    float* Bg = B2 + TM * LDH;
    stage_matrix(sW1, LDW1, a.W1, HID, IN, HID, tid);
    stage_matrix(sW2, LDW, a.W2, HID, HID, HID, tid);
    stage_matrix(sWo, LDW, a.Wo, OUT, HID, 16, tid);
    if (tid < 64) { sB[tid] = a.b1[tid]; sB[64 + tid] = a.b2[tid]; }
    if (tid < 16) sB[128 + tid] = (tid < OUT) ? a.bo[tid] : 0.0f;
    __syncthreads();

    f32x16 accW1[2][NT1], accW2[2][2];
    f32x4 accO[4];
#pragma unroll
    for (int m = 0; m < 2; m++) { zero_acc(accW1[m]); zero_acc(accW2[m]); }
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) accO[nt][r] = 0.0f;
    float cs1 = 0.0f, cs2 = 0.0f, cso = 0.0f;      // bias gradients: lane = column

    const int64_t n_tiles = (a.n + TM - 1) / TM;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t p0 = tile * TM;
        mlp::assemble<KIND, IN, TM>(B0, LD0, a.segs, p0, a.n, lane);
        f32x16 acc[2];
        unsigned m1 = 0u, m2 = 0u;
        // ---- forward ----
        zero_acc(acc);
        gemm_xwT<IN_PAD / 2>(B0, LD0, sW1, LDW1, acc, lane);
        {
            int bit = 0;
            ACC2_FOREACH(nt, r, row, col, lane) {
                const float v = fmaxf(acc[nt][r] + sB[col], 0.0f);
                B1[row * LDH + col] = v;
                if (v > 0.0f) m1 |= 1u << bit;
                bit++;
            }
        }
        zero_acc(acc);
        gemm_xwT<HID / 2>(B1, LDH, sW2, LDW, acc, lane);
        {
            int bit = 0;
            ACC2_FOREACH(nt, r, row, col, lane) {
                const float v = fmaxf(acc[nt][r] + sB[64 + col], 0.0f);
                B2[row * LDH + col] = v;
                if (v > 0.0f) m2 |= 1u << bit;
                bit++;
            }
        }
        // output layer -> G3 = g_y * y (1 - y)  (rows of points beyond n: 0)
        {
            const int l15 = lane & 15, l4 = lane >> 4;
            f32x4 o[2];
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) o[m][r] = 0.0f;
#pragma unroll 4
            for (int kk = 0; kk < HID / 4; kk++) {
                const int k = 4 * kk + l4;
                const float b = sWo[l15 * LDW + k];
#pragma unroll
                for (int m = 0; m < 2; m++)
                    o[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(B2[(16 * m + l15) * LDH + k], b, o[m], 0, 0, 0);
            }
            const float bias = sB[128 + l15];
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * m + l4 * 4 + r;
                    const int64_t p = p0 + row;
                    float g = 0.0f;
                    if (l15 < OUT && p < a.n) {
                        const float y = 1.0f / (1.0f + __expf(-(o[m][r] + bias)));
                        g = a.g_y[p * OUT + l15] * y * (1.0f - y);
                    }
                    Bg[row * LDG + l15] = g;
                }
        }
        // ---- backward ----
        wgrad_out_acc(Bg, LDG, B2, LDH, accO, lane);                      // dWo += G3^T A2
        if (lane < 16) cso += col_sum(Bg, LDG, lane);
        zero_acc(acc);
        gemm_xw<8>(Bg, LDG, sWo, LDW, 0, HID, acc, lane);                 // G3 [32x16] . Wo [16x64]
        {
            int bit = 0;
            ACC2_FOREACH(nt, r, row, col, lane) {
                B2[row * LDH + col] = ((m2 >> bit) & 1u) ? acc[nt][r] : 0.0f;      // G2 over A2 (dWo is done with it)
                bit++;
            }
        }
        wgrad_acc<2>(B2, LDH, B1, LDH, HID, accW2, lane);                 // dW2 += G2^T A1
        cs2 += col_sum(B2, LDH, lane);
        zero_acc(acc);
        gemm_xw<HID / 2>(B2, LDH, sW2, LDW, 0, HID, acc, lane);           // G2 . W2
        {
            int bit = 0;
            ACC2_FOREACH(nt, r, row, col, lane) {
                B1[row * LDH + col] = ((m1 >> bit) & 1u) ? acc[nt][r] : 0.0f;      // G1 over A1
                bit++;
            }
        }
        wgrad_acc<NT1>(B1, LDH, B0, LD0, IN, accW1, lane);                // dW1 += G1^T X
        cs1 += col_sum(B1, LDH, lane);
        if (a.g_x) {                                                      // g_x = G1 . W1, 64 columns at a time
#pragma unroll
            for (int c0 = 0; c0 < IN; c0 += 64) {
                zero_acc(acc);
                gemm_xw<HID / 2>(B1, LDH, sW1, LDW1, c0, IN, acc, lane);
                ACC2_FOREACH(nt, r, row, col, lane) {
                    const int64_t p = p0 + row;
                    if (p < a.n && c0 + col < IN) a.g_x[p * a.gx_stride + c0 + col] = acc[nt][r];
                }
            }
        }
    }
    // ---- reduce the weight gradients over the workgroup, add to global ----
    float* sRed = sTiles;
    flush_w<NT1>(sRed, accW1, a.dW1, HID, IN, IN, tid);
    flush_w<2>(sRed, accW2, a.dW2, HID, HID, HID, tid);
    flush_o(sRed, accO, a.dWo, OUT, tid);
    if (cs1 != 0.0f) unsafeAtomicAdd(a.db1 + lane, cs1);
    if (cs2 != 0.0f) unsafeAtomicAdd(a.db2 + lane, cs2);
    if (lane < OUT && cso != 0.0f) unsafeAtomicAdd(a.dbo + lane, cso);
}

template <int KIND, int IN, int OUT>
int launch_train2(const Train2Args& a, hipStream_t s)
{
    constexpr int IN_PAD = (IN + 1) / 2 * 2;
    constexpr int LDW1 = IN_PAD + 1, LDW = HID + 1;
    constexpr int PER_WAVE = TM * ((IN_PAD + 1) + 2 * (HID + 1) + 17);
    constexpr int TILES = WAVES * PER_WAVE > 64 * LDR ? WAVES * PER_WAVE : 64 * LDR;
    constexpr size_t lds = sizeof(float) * (HID * LDW1 + HID * LDW + 16 * LDW + 144 + TILES);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = mlp2_train_kernel<KIND, IN, OUT>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int64_t n_tiles = (a.n + TM - 1) / TM;
    int grid = (int)((n_tiles + WAVES - 1) / WAVES);
    if (grid > 256) grid = 256;
    kern<<<grid, THREADS, lds, s>>>(a);
    return ia::check_launch("ia_mlp_bwd_fused");
}

// ------------------------------------------------------------------------------------------------
struct SdfTrainArgs {
    int64_t n;
    int n_segs;
    Seg segs[MAX_SEGS];
    const float *W1, *b1, *Wo, *bo;
    const float* jac;      // [n,32,3]
    const float* g_out;    // [n,13]
    const float* q;        // [n,3]
    float *gE, *gG;        // [n,32] each
    float* gXYZ;           // [n,3] or NULL: first-order d L / d (2 x' - 1) (the xyz columns of dz W1), for d L / d x
    float *dW1, *db1, *dWo, *dbo;      // [64,35] [64] [13,64] [13]  accumulated into
};

template <bool WANT_XYZ>
__global__ __launch_bounds__(THREADS) void sdf_train_kernel(SdfTrainArgs a)
{
    constexpr int IN = 35, IN_PAD = 36, OUT = 13;
    constexpr int LDW1 = IN_PAD + 1, LDW = HID + 1;
    constexpr int LDI = IN_PAD + 1, LDH = HID + 1, LDG = 17;
    constexpr int PER_WAVE = TM * (2 * LDI + 2 * LDH + LDG);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;
    float* sWo = sW1 + HID * LDW1;
    float* sB = sWo + 16 * LDW;
    float* sTiles = sB + 144;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* T0 = sTiles + wave * PER_WAVE;      // H  (input rows)        [32][37]
    float* T3 = T0 + TM * LDI;                 // U  (JVP of the input)  [32][37]
    float* T1 = T3 + TM * LDI;                 // A, then DZ             [32][65]
    float* T2 = T1 + TM * LDH;                 // GZ                     [32][65]
    float* Tg = T2 + TM * LDH;                 // g_out                  [32][17]
    stage_matrix(sW1, LDW1, a.W1, HID, IN, HID, tid);
    stage_matrix(sWo, LDW, a.Wo, OUT, HID, 16, tid);
    if (tid < 64) sB[tid] = a.b1[tid];
    if (tid < 16) sB[128 + tid] = (tid < OUT) ? a.bo[tid] : 0.0f;
    __syncthreads();

    f32x16 accW1[2][2];
    f32x4 accO[4];
#pragma unroll
    for (int m = 0; m < 2; m++) zero_acc(accW1[m]);
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) accO[nt][r] = 0.0f;
    float cs1 = 0.0f, cso = 0.0f;
    float dgs[2] = {0.0f, 0.0f};               // sum over points of dgz * s, column 32 nt + lr (both lk halves)

    const int64_t n_tiles = (a.n + TM - 1) / TM;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t p0 = tile * TM;
        f32x16 acc[2];
        float sig[2][16];
        // ---- forward: h -> z -> (a, s) ----
        mlp::assemble<0, IN, TM>(T0, LDI, a.segs, p0, a.n, lane);
        zero_acc(acc);
        gemm_xwT<IN_PAD / 2>(T0, LDI, sW1, LDW1, acc, lane);
        ACC2_FOREACH(nt, r, row, col, lane) {
            const float z = acc[nt][r] + sB[col];
            float sg;
            T1[row * LDH + col] = mlp::softplus100(z, sg);
            sig[nt][r] = sg;
        }
        // ---- g_out tile; dWo += g_out^T A ----
        for (int i = lane; i < TM * 16; i += 64) {
            const int r = i >> 4, c = i & 15;
            const int64_t p = p0 + r;
            Tg[r * LDG + c] = (c < OUT && p < a.n) ? a.g_out[p * OUT + c] : 0.0f;
        }
        wgrad_out_acc(Tg, LDG, T1, LDH, accO, lane);
        if (lane < 16) cso += col_sum(Tg, LDG, lane);
        // ---- gz = s * W2[0,:] -> T2 ; gh = gz W1 -> gG ----
        ACC2_FOREACH(nt, r, row, col, lane) T2[row * LDH + col] = sig[nt][r] * sWo[col];
        zero_acc(acc);
        gemm_xw<HID / 2>(T2, LDH, sW1, LDW1, 0, IN_PAD, acc, lane);
        ACC2_FOREACH(nt, r, row, col, lane) {
            const int64_t p = p0 + row;
            if (col < 32 && p < a.n) a.gG[p * 32 + col] = acc[nt][r];
        }
        // ---- u = [J q | 2 q] per point (lanes 0..31 = points) -> T3 ; dW1 += GZ^T U ----
        if (lane < TM) {
            const int64_t p = p0 + lane;
            float* urow = T3 + lane * LDI;
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
        wgrad_acc<2>(T2, LDH, T3, LDI, IN, accW1, lane);
        // ---- dgz = u W1^T ; dz2 = dgz * W2[0] * 100 s (1 - s) ; sum dgz * s ----
        zero_acc(acc);
        gemm_xwT<IN_PAD / 2>(T3, LDI, sW1, LDW1, acc, lane);
        float dz2[2][16];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float s = sig[nt][r], d = acc[nt][r];
                dz2[nt][r] = d * sWo[32 * nt + (lane & 31)] * 100.0f * s * (1.0f - s);
                dgs[nt] += d * s;
            }
        // ---- da = g_out W2 ; dz = da * s + dz2 -> T1 (A is no longer needed) ----
        zero_acc(acc);
        gemm_xw<8>(Tg, LDG, sWo, LDW, 0, HID, acc, lane);
        ACC2_FOREACH(nt, r, row, col, lane) T1[row * LDH + col] = acc[nt][r] * sig[nt][r] + dz2[nt][r];
        wgrad_acc<2>(T1, LDH, T0, LDI, IN, accW1, lane);                  // dW1 += DZ^T H
        cs1 += col_sum(T1, LDH, lane);
        // ---- gE = (dz W1)[:, :32] ----
        zero_acc(acc);
        gemm_xw<HID / 2>(T1, LDH, sW1, LDW1, 0, IN_PAD, acc, lane);
        ACC2_FOREACH(nt, r, row, col, lane) {
            const int64_t p = p0 + row;
            if (col < 32 && p < a.n) a.gE[p * 32 + col] = acc[nt][r];
            if (WANT_XYZ && col >= 32 && col < 35 && p < a.n) a.gXYZ[p * 3 + (col - 32)] = acc[nt][r];
        }
    }
    float* sRed = sTiles;
    flush_w<2>(sRed, accW1, a.dW1, HID, IN, IN, tid);
    flush_o(sRed, accO, a.dWo, OUT, tid);
    if (cs1 != 0.0f) unsafeAtomicAdd(a.db1 + lane, cs1);
    if (lane < OUT && cso != 0.0f) unsafeAtomicAdd(a.dbo + lane, cso);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
        if (dgs[nt] != 0.0f) unsafeAtomicAdd(a.dWo + 32 * nt + (lane & 31), dgs[nt]);      // row 0 of dWo
}

}  // namespace

IA_EXPORT int ia_mlp_bwd_fused(int kind, int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
                               const int* seg_width, const float* seg_mul, const float* seg_add, const float* W1,
                               const float* b1, const float* W2, const float* b2, const float* Wo, const float* bo,
                               const float* g_y, float* g_x, int gx_stride, float* dW1, float* db1, float* dW2,
                               float* db2, float* dWo, float* dbo, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(kind == 1 || kind == 2, "ia_mlp_bwd_fused: kind must be 1 (radiance) or 2 (material)");
    IA_REQUIRE(dW1 && db1 && dW2 && db2 && dWo && dbo, "ia_mlp_bwd_fused: all six gradient buffers are required");
    Train2Args a = {};
    a.n = n; a.n_segs = n_segs;
    int r = mlp::fill_segs(a.segs, kind, n_segs, seg_ptr, seg_stride, seg_width, seg_mul, seg_add);
    if (r != IA_OK) return r;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Wo = Wo; a.bo = bo;
    a.g_y = g_y; a.g_x = g_x; a.gx_stride = gx_stride;
    a.dW1 = dW1; a.db1 = db1; a.dW2 = dW2; a.db2 = db2; a.dWo = dWo; a.dbo = dbo;
    return kind == 1 ? launch_train2<1, 67, 3>(a, (hipStream_t)stream) : launch_train2<2, 48, 5>(a, (hipStream_t)stream);
}

IA_EXPORT int ia_sdf_mlp_bwd_fused(int64_t n, int n_segs, const float* const* seg_ptr, const int* seg_stride,
                                   const int* seg_width, const float* seg_mul, const float* seg_add, const float* W1,
                                   const float* b1, const float* Wo, const float* bo, const float* jac,
                                   const float* g_out, const float* q, float* gE, float* gG, float* dW1, float* db1,
                                   float* dWo, float* dbo, float* g_xyz, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(dW1 && db1 && dWo && dbo && gE && gG, "ia_sdf_mlp_bwd_fused: all output buffers are required");
    SdfTrainArgs a = {};
    a.n = n; a.n_segs = n_segs;
    int r = mlp::fill_segs(a.segs, 0, n_segs, seg_ptr, seg_stride, seg_width, seg_mul, seg_add);
    if (r != IA_OK) return r;
    a.W1 = W1; a.b1 = b1; a.Wo = Wo; a.bo = bo; a.jac = jac; a.g_out = g_out; a.q = q;
    a.gE = gE; a.gG = gG; a.dW1 = dW1; a.db1 = db1; a.dWo = dWo; a.dbo = dbo; a.gXYZ = g_xyz;
    constexpr int PER_WAVE = TM * (2 * 37 + 2 * 65 + 17);
    constexpr int TILES = WAVES * PER_WAVE > 64 * LDR ? WAVES * PER_WAVE : 64 * LDR;
    constexpr size_t lds = sizeof(float) * (HID * 37 + 16 * 65 + 144 + TILES);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)sdf_train_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)sdf_train_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int64_t n_tiles = (n + TM - 1) / TM;
    int grid = (int)((n_tiles + WAVES - 1) / WAVES);
    if (grid > 256) grid = 256;
    if (g_xyz) sdf_train_kernel<true><<<grid, THREADS, lds, (hipStream_t)stream>>>(a);
    else sdf_train_kernel<false><<<grid, THREADS, lds, (hipStream_t)stream>>>(a);
    return ia::check_launch("ia_sdf_mlp_bwd_fused");
}
