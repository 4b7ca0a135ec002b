// mlp_tile.h -- shared pieces of the fused MLP kernels (mlp.hip, mlp_bwd.hip): input-row assembly in LDS.
#pragma once
#include "ia_common.h"

namespace mlp {

constexpr int MAX_SEGS = 5;

struct Seg {
    const float* p;
    int stride;   // floats between consecutive points
    int width;    // columns taken from this source
    float mul, add;
};

// compile-time segment widths per MLP kind (0 SDF, 1 radiance, 2 material): lets the compiler turn the
// index arithmetic into shifts / multiply-high and issue all loads of a segment back to back
template <int KIND> struct SegWidths;
template <> struct SegWidths<0> { static constexpr int n = 2; static constexpr int w[MAX_SEGS] = {32, 3, 0, 0, 0}; };
template <> struct SegWidths<1> { static constexpr int n = 5; static constexpr int w[MAX_SEGS] = {32, 3, 13, 16, 3}; };
template <> struct SegWidths<2> { static constexpr int n = 3; static constexpr int w[MAX_SEGS] = {32, 3, 13, 0, 0}; };
// kind 3: the SDF head fed straight from the level-major result of ia_hashgrid_fwd_xcd (segment 0 = float2 [16][n],
// segs[0].stride = n), no [n,32] row in HBM
template <> struct SegWidths<3> { static constexpr int n = 2; static constexpr int w[MAX_SEGS] = {32, 3, 0, 0, 0}; };

// Two-phase segment transfer: all global loads of a tile are issued into registers first (SegRegs), the LDS stores
// follow -- one exposed global-load latency per tile instead of one per segment.
template <int WIDTH, int ROWS>
struct SegRegs {
    static constexpr int NV = (WIDTH % 4 == 0) ? (ROWS * (WIDTH / 4) + 63) / 64 : 0;     // float4 pieces per lane
    static constexpr int NS = (ROWS * WIDTH + 63) / 64;                                  // scalars per lane
    float4 q[NV > 0 ? NV : 1];
    float v[NS > 0 ? NS : 1];
    bool vec;
};

template <int WIDTH, int ROWS>
__device__ __forceinline__ void seg_load(SegRegs<WIDTH, ROWS>& r, const Seg& sg, int64_t p0, int64_t n, int lane)
{
    if constexpr (WIDTH == 0) return;
    r.vec = false;
    if constexpr (WIDTH % 4 == 0) {
        // 16-byte loads when the rows allow it (hash features: 32 wide, SH: 16 wide)
        r.vec = ((sg.stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(sg.p) & 15) == 0);
        if (r.vec) {
            constexpr int V = WIDTH / 4;
#pragma unroll
            for (int j = 0; j < SegRegs<WIDTH, ROWS>::NV; j++) {
                const int i = j * 64 + lane;
                const int row = i / V, c4 = i % V;
                const int64_t p = p0 + row;
                r.q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < ROWS * V && p < n) r.q[j] = *reinterpret_cast<const float4*>(sg.p + p * sg.stride + c4 * 4);
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < SegRegs<WIDTH, ROWS>::NS; j++) {
        const int i = j * 64 + lane;
        const int row = i / WIDTH, c = i % WIDTH;
        const int64_t p = p0 + row;
        r.v[j] = (i < ROWS * WIDTH && p < n) ? sg.p[p * sg.stride + c] : 0.0f;
    }
}

template <int WIDTH, int ROWS>
__device__ __forceinline__ void seg_store(const SegRegs<WIDTH, ROWS>& r, float* sT, int ldx, int col0, const Seg& sg,
                                          int64_t p0, int64_t n, int lane)
{
    if constexpr (WIDTH == 0) return;
    if constexpr (WIDTH % 4 == 0) {
        if (r.vec) {
            constexpr int V = WIDTH / 4;
#pragma unroll
            for (int j = 0; j < SegRegs<WIDTH, ROWS>::NV; j++) {
                const int i = j * 64 + lane;
                if (i >= ROWS * V) break;
                const int row = i / V, c4 = i % V;
                const bool ok = p0 + row < n;           // rows beyond n: zeros (NOT add), as the one-phase path did
                float* d = sT + row * ldx + col0 + c4 * 4;
                d[0] = ok ? r.q[j].x * sg.mul + sg.add : 0.0f; d[1] = ok ? r.q[j].y * sg.mul + sg.add : 0.0f;
                d[2] = ok ? r.q[j].z * sg.mul + sg.add : 0.0f; d[3] = ok ? r.q[j].w * sg.mul + sg.add : 0.0f;
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < SegRegs<WIDTH, ROWS>::NS; j++) {
        const int i = j * 64 + lane;
        if (i >= ROWS * WIDTH) break;
        const int row = i / WIDTH, c = i % WIDTH;
        sT[row * ldx + col0 + c] = (p0 + row < n) ? r.v[j] * sg.mul + sg.add : 0.0f;
    }
}

template <int KIND, int IN, int ROWS = 64>
__device__ __forceinline__ void assemble(float* sT, int ldx, const Seg* segs, int64_t p0, int64_t n, int lane)
{
    using SW = SegWidths<KIND>;
    constexpr int IN_PAD = (IN + 1) / 2 * 2;
    if constexpr (KIND == 3) {
        // 16 levels x ROWS points of float2, level-major: lane -> (level, row) so that a half-wave reads 32 consecutive
        // float2 (256 contiguous bytes) of ONE level; all loads in flight before the LDS stores
        static_assert(ROWS == 32, "level-major assembly is written for 32-point tiles");
        const float2* lv = reinterpret_cast<const float2*>(segs[0].p);
        const int64_t ls = segs[0].stride;                  // points per level (= n)
        float2 q[8];
        SegRegs<3, ROWS> r1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = j * 64 + lane, lvl = i >> 5, row = i & 31;
            q[j] = (p0 + row < n) ? lv[(int64_t)lvl * ls + p0 + row] : make_float2(0.f, 0.f);
        }
        seg_load(r1, segs[1], p0, n, lane);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = j * 64 + lane, lvl = i >> 5, row = i & 31;
            sT[row * ldx + 2 * lvl] = q[j].x; sT[row * ldx + 2 * lvl + 1] = q[j].y;
        }
        seg_store(r1, sT, ldx, 32, segs[1], p0, n, lane);
        if (IN_PAD > IN && lane < ROWS) sT[lane * ldx + IN] = 0.0f;
        return;
    }
    SegRegs<SW::w[0], ROWS> r0; SegRegs<SW::w[1], ROWS> r1; SegRegs<SW::w[2], ROWS> r2;
    SegRegs<SW::w[3], ROWS> r3; SegRegs<SW::w[4], ROWS> r4;
    seg_load(r0, segs[0], p0, n, lane); seg_load(r1, segs[1], p0, n, lane); seg_load(r2, segs[2], p0, n, lane);
    seg_load(r3, segs[3], p0, n, lane); seg_load(r4, segs[4], p0, n, lane);
    seg_store(r0, sT, ldx, 0, segs[0], p0, n, lane);
    seg_store(r1, sT, ldx, SW::w[0], segs[1], p0, n, lane);
    seg_store(r2, sT, ldx, SW::w[0] + SW::w[1], segs[2], p0, n, lane);
    seg_store(r3, sT, ldx, SW::w[0] + SW::w[1] + SW::w[2], segs[3], p0, n, lane);
    seg_store(r4, sT, ldx, SW::w[0] + SW::w[1] + SW::w[2] + SW::w[3], segs[4], p0, n, lane);
    if (IN_PAD > IN && lane < ROWS) sT[lane * ldx + IN] = 0.0f;
}

inline int fill_segs(Seg* segs, int kind, int n_segs, const float* const* seg_ptr, const int* seg_stride,
                     const int* seg_width, const float* seg_mul, const float* seg_add)
{
    static const int W[3][MAX_SEGS] = {{32, 3, 0, 0, 0}, {32, 3, 13, 16, 3}, {32, 3, 13, 0, 0}};
    static const int NS[3] = {2, 5, 3};
    if (kind < 0 || kind > 2) { ia::set_error("unknown MLP kind"); return IA_ERR_INVALID; }
    if (n_segs != NS[kind]) { ia::set_error("MLP kind %d takes exactly %d input segments", kind, NS[kind]); return IA_ERR_INVALID; }
    for (int s = 0; s < MAX_SEGS; s++) { segs[s].p = nullptr; segs[s].stride = 0; segs[s].width = 0; segs[s].mul = 1.f; segs[s].add = 0.f; }
    for (int s = 0; s < n_segs; s++) {
        if (seg_width[s] != W[kind][s]) { ia::set_error("MLP kind %d: segment %d must be %d wide", kind, s, W[kind][s]); return IA_ERR_INVALID; }
        segs[s].p = seg_ptr[s];
        segs[s].stride = seg_stride[s];
        segs[s].width = seg_width[s];
        segs[s].mul = seg_mul ? seg_mul[s] : 1.0f;
        segs[s].add = seg_add ? seg_add[s] : 0.0f;
    }
    return IA_OK;
}

// Softplus(beta=100, threshold=20) and its derivative with the hardware exp/log units
// Straight-line: the raw v_exp_f32 / v_log_f32 (base 2) with the scale factors applied here, and both log1p forms computed and
// selected.  (__expf / __logf wrap the same instructions in range fix-ups for denormal results / arguments -- compare, ldexp,
// select -- and the conditional around the logarithm compiled to a divergent branch per activation: 60 branch regions and
// ~190 of the ~1180 VALU instructions of a 32-point tile of the SDF head.  1 + e is in (1, 2], and an e below the normal range
// contributes nothing to max(bx, 0) + l at the precision of the result.)
__device__ __forceinline__ float softplus100(float z, float& sig)
{
    const float bx = 100.0f * z;
    const float e = __builtin_amdgcn_exp2f(-fabsf(bx) * 1.44269504088896340736f);      // exp(-|bx|) in [0, 1]
    const float lg = __builtin_amdgcn_logf(1.0f + e) * 0.69314718055994530942f;       // log(1 + e)
    const float sm = e * (1.0f - 0.5f * e);
    const float l = (e < 1e-4f) ? sm : lg;                                              // log1p(e)
    sig = bx >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
    return bx > 20.0f ? z : (fmaxf(bx, 0.0f) + l) * 0.01f;
}

}  // namespace mlp
