// deform.hip -- candidate bookkeeping of the SNARF deformer + per-sample shading prep for gfx950.
// Replaces the boolean-mask gathers / scatters / torch.gather chains of
//   SNARFDeformer.deform          models/deformers/snarf_deformer.py:187-261
//   ForwardDeformer.forward       models/deformers/fast_snarf/deformer_torch.py:35-55 (eval branch)
//   rgb_normal_alpha_fn prologue  models/intrinsic_avatar.py:1032-1064 (positions, normals, reflected dirs)
//   get_alpha + Laplace density   models/intrinsic_avatar.py:390-394, models/rf/density.py:25-30
// with four small kernels and no host-side nonzero():
//   1. filter (K9) fused with the per-point valid-candidate count,
//   2. exclusive scan (core.hip) -> packed candidate list [Q] in (point, init) order (deterministic),
//   3. the SDF network runs on the packed list (mlp.hip / hashgrid.hip),
//   4. select: first-minimum SDF over each point's candidates, gather of position / feature /
//      canonical gradient, push-forward of the gradient with the blended bone rotation.
#include "ia_common.h"

namespace {

constexpr int THREADS = 256;

// ---- 1. filter + count ---------------------------------------------------------
// The candidates of a workgroup's 256 points are one contiguous 40 KB block: it is staged into LDS with coalesced
// 16-byte loads (a lane-per-point read of 12-byte pieces at a 156-byte stride fetched ~5x the bytes from the fabric),
// the O(I^2) pair test then runs out of LDS (row stride 39 words: conflict-free).  Same arithmetic as before.
constexpr int FC_MAX_I = 16;
__global__ __launch_bounds__(THREADS) void filter_count_kernel(int64_t P, int I, const float* __restrict__ x,
                                                                const uint8_t* __restrict__ valid,
                                                                uint8_t* __restrict__ mask, int32_t* __restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) float s_x[];       // [THREADS][I*3]
    uint8_t* s_v = reinterpret_cast<uint8_t*>(s_x + THREADS * I * 3);  // [THREADS][I]
    const int64_t p0 = (int64_t)blockIdx.x * THREADS;
    const int rows = (int)((P - p0) < THREADS ? (P - p0) : THREADS);
    const int nf = rows * I * 3, nb = rows * I;
    const float* gx = x + p0 * I * 3;
    const uint8_t* gv = valid + p0 * I;
    if ((reinterpret_cast<uintptr_t>(gx) & 15) == 0) {
        const float4* g4 = reinterpret_cast<const float4*>(gx);
        float4* s4 = reinterpret_cast<float4*>(s_x);
        for (int i = threadIdx.x; i < nf / 4; i += THREADS) s4[i] = g4[i];
        for (int i = (nf / 4) * 4 + threadIdx.x; i < nf; i += THREADS) s_x[i] = gx[i];
    } else {
        for (int i = threadIdx.x; i < nf; i += THREADS) s_x[i] = gx[i];
    }
    for (int i = threadIdx.x; i < nb; i += THREADS) s_v[i] = gv[i];
    __syncthreads();
    const int t = threadIdx.x;
    const bool live = t < rows;
    const float* xr = s_x + t * I * 3;
    const uint8_t* vr = s_v + t * I;
    int c = 0;
    unsigned keep_bits = 0;
    for (int i = 0; live && i < I; i++) {
        bool keep = vr[i] != 0;
        if (keep) {
            const float xi0 = xr[i * 3 + 0], xi1 = xr[i * 3 + 1], xi2 = xr[i * 3 + 2];
            for (int j = i + 1; j < I; j++) {
                if (!vr[j]) continue;
                const float d0 = xi0 - xr[j * 3 + 0];
                const float d1 = xi1 - xr[j * 3 + 1];
                const float d2 = xi2 - xr[j * 3 + 2];
                const float dist = d0 * d0 + d1 * d1 + d2 * d2;
                if ((double)dist < 0.0001 * 0.0001) { keep = false; break; }
            }
        }
        keep_bits |= (keep ? 1u : 0u) << i;
        c += keep ? 1 : 0;
    }
    if (live) cnt[p0 + t] = c;
    // mask rows back through LDS so that the byte stores are coalesced too
    __syncthreads();
    for (int i = 0; live && i < I; i++) s_v[t * I + i] = (keep_bits >> i) & 1u;
    __syncthreads();
    uint8_t* gm = mask + p0 * I;
    for (int i = threadIdx.x; i < nb; i += THREADS) gm[i] = s_v[i];
}

// ---- 2. packed candidate list --------------------------------------------------
__global__ __launch_bounds__(THREADS) void compact_fill_kernel(int64_t P, int I, const float* __restrict__ x,
                                                                const uint8_t* __restrict__ mask,
                                                                const int32_t* __restrict__ start,
                                                                float* __restrict__ cand_x, int32_t* __restrict__ cand_src)
{
    const int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= P) return;
    int q = start[p];
    for (int i = 0; i < I; i++) {
        if (!mask[p * I + i]) continue;
        cand_x[(int64_t)q * 3 + 0] = x[(p * I + i) * 3 + 0];
        cand_x[(int64_t)q * 3 + 1] = x[(p * I + i) * 3 + 1];
        cand_x[(int64_t)q * 3 + 2] = x[(p * I + i) * 3 + 2];
        cand_src[q] = (int32_t)(p * I + i);
        q++;
    }
}

// ---- 1 + 2 in ONE pass: filter, count, chained scan, compaction ---------------------
// filter_count -> exclusive scan -> compact_fill read x twice and the mask once more (338 B per point over three kernels).
// Here a tile of FCC_ROWS points is staged into LDS once, filtered, counted, and its candidates are written straight to their
// final positions: the tile's global offset comes from a decoupled look-back over the tile descriptors (status | value in
// one 64-bit word).
//  * cand_x MAY ALIAS x (in-place compaction): a tile only learns its offset after every earlier tile has published its
//    count, i.e. after those tiles finished reading their rows, and its own output ends at or before the end of its own rows.
//  * The descriptor accesses are RELAXED device-scope atomics on purpose: an acquire / release at device scope invalidates /
//    writes back the XCD's whole L2 on this part (8 non-coherent L2s), once per tile -- measured 9x slower than the three
//    kernels.  No fence is needed: a tile's count depends on the rows it loaded (so those loads are complete when the count
//    is published), nothing written by this kernel is read back by it, and tiles are whole 128-byte lines.
//  * tile = workgroup index: workgroups are dispatched in index order (per XCD as well), so the lowest unfinished tile is
//    always resident and every wait ends.  A ticket counter instead costs one same-address device-scope atomic per tile, 30 ns
//    each, serialised = 52 ms per step.  One poll covers FCC_LOOK x 64 predecessors (a 64-tile window ran at 18 ns per tile).
//  * Measured alternatives that were slower (tools/filter_probe.py, 49 M points): 128-row tiles with a 512-tile window (5.4 ms
//    against 4.6 ms: the look-back cost is per tile), persistent workgroups that load the next tile's rows into registers
//    before the look-back of the current one (6.1 ms: 40 more live registers spill in the pair tests).  Loads alone: 2.2 ms.
constexpr int FCC_ROWS = 256, FCC_LOOK = 4;
constexpr uint64_t FCC_AGGREGATE = 1ull << 32, FCC_PREFIX = 2ull << 32;

template <int IT, bool LOCAL>
__global__ __launch_bounds__(FCC_ROWS) void filter_compact_kernel(int64_t P, int I_rt, uint32_t magic_I, int n_tiles, const float* x,
                                                                  const uint8_t* __restrict__ valid, uint64_t* desc,
                                                                  int32_t* __restrict__ cnt, int32_t* __restrict__ start, float* cand_x,
                                                                  int32_t* __restrict__ cand_src, uint8_t* __restrict__ mask,
                                                                  int32_t* __restrict__ total)
{
    const int I = IT > 0 ? IT : I_rt;
    constexpr int U = IT > 0 ? (IT * 3 + 3) / 4 : (FC_MAX_I * 3) / 4;   // float4 per thread and tile
    constexpr int UV = 4;                                              // valid words per thread and tile (I <= 16)
    extern __shared__ __attribute__((aligned(16))) float s_x[];         // [ROWS][I*3]
    uint8_t* s_v = reinterpret_cast<uint8_t*>(s_x + FCC_ROWS * I * 3);  // [ROWS][I]   (4-byte aligned)
    __shared__ int s_lo[FCC_ROWS];
    __shared__ unsigned short s_kb[FCC_ROWS];
    __shared__ int s_wsum[FCC_ROWS / 64];
    __shared__ int s_prefix;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // tile strides are multiples of 16 bytes (x) and 4 bytes (valid): alignment is a property of the launch
    const bool wide = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(valid) & 3) == 0;
    float4 r[U];
    uint32_t vw[UV];
    auto tile_rows = [&](int tile) { const int64_t left = P - (int64_t)tile * FCC_ROWS; return (int)(left < FCC_ROWS ? left : FCC_ROWS); };
    auto issue_loads = [&](int tile) {          // clamped, unconditional: the registers stay registers
        const int rows = tile_rows(tile);
        const int n4 = rows * I * 3 / 4, nw = rows * I / 4;
        const float4* g4 = reinterpret_cast<const float4*>(x + (int64_t)tile * FCC_ROWS * I * 3);
        const uint32_t* gw = reinterpret_cast<const uint32_t*>(valid + (int64_t)tile * FCC_ROWS * I);
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = n4 > 0 ? g4[min(t + u * FCC_ROWS, n4 - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < UV; u++) vw[u] = nw > 0 ? gw[min(t + u * FCC_ROWS, nw - 1)] : 0u;
    };
    auto stage = [&](int tile) {                // registers (or, unaligned, global memory) -> LDS
        const int rows = tile_rows(tile);
        const int nf = rows * I * 3, nb = rows * I;
        const float* gx = x + (int64_t)tile * FCC_ROWS * I * 3;
        const uint8_t* gv = valid + (int64_t)tile * FCC_ROWS * I;
        const int n4 = wide ? nf / 4 : 0, nw = wide ? nb / 4 : 0;
        float4* s4 = reinterpret_cast<float4*>(s_x);
        uint32_t* sw = reinterpret_cast<uint32_t*>(s_v);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (t + u * FCC_ROWS < n4) s4[t + u * FCC_ROWS] = r[u];
#pragma unroll
        for (int u = 0; u < UV; u++)
            if (t + u * FCC_ROWS < nw) sw[t + u * FCC_ROWS] = vw[u];
        for (int i = n4 * 4 + t; i < nf; i += FCC_ROWS) s_x[i] = gx[i];
        for (int i = nw * 4 + t; i < nb; i += FCC_ROWS) s_v[i] = gv[i];
    };
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {      // the host launches one workgroup per tile
        if (wide) issue_loads(tile);
        stage(tile);
        __syncthreads();
        const int64_t p0 = (int64_t)tile * FCC_ROWS;
        const int rows = tile_rows(tile);
        const int nb = rows * I;
        // K9 (filter.cu:10-77): a valid candidate is dropped when a LATER valid one lies within 1e-4; same arithmetic as filter_kernel
        unsigned keep_bits = 0;
        if (t < rows) {
            const float* xr = s_x + t * I * 3;
            unsigned vbits = 0;
            for (int i = 0; i < I; i++) vbits |= (s_v[t * I + i] ? 1u : 0u) << i;
            if constexpr (IT > 0) {
                // straight line: the row goes to registers with its LDS reads in flight together, then all IT (IT - 1) / 2 pair
                // tests without a branch (the data-dependent loop below is a chain of ~15 dependent LDS round trips per point).
                // (double)dist < 0.0001 * 0.0001 for a float dist <=> dist < the smallest float above that double (0x322bcc78).
                const float thr = __uint_as_float(0x322bcc78u);
                float xv[IT * 3];
#pragma unroll
                for (int k = 0; k < IT * 3; k++) xv[k] = xr[k];
                unsigned kill = 0;                           // bit i: a LATER valid candidate lies within the threshold
#pragma unroll
                for (int i = 0; i < IT; i++) {
#pragma unroll
                    for (int j = i + 1; j < IT; j++) {
                        const float d0 = xv[i * 3 + 0] - xv[j * 3 + 0];
                        const float d1 = xv[i * 3 + 1] - xv[j * 3 + 1];
                        const float d2 = xv[i * 3 + 2] - xv[j * 3 + 2];
                        const float dist = d0 * d0 + d1 * d1 + d2 * d2;
                        kill |= ((dist < thr) && ((vbits >> j) & 1u)) ? (1u << i) : 0u;
                    }
                }
                keep_bits = vbits & ~kill;
            } else {
                for (int i = 0; i < I; i++) {
                    if (!((vbits >> i) & 1u)) continue;
                    const float xi0 = xr[i * 3 + 0], xi1 = xr[i * 3 + 1], xi2 = xr[i * 3 + 2];
                    bool keep = true;
                    for (unsigned rest = vbits >> (i + 1), j = i + 1; rest; rest >>= 1, j++) {
                        if (!(rest & 1u)) continue;
                        const float d0 = xi0 - xr[j * 3 + 0];
                        const float d1 = xi1 - xr[j * 3 + 1];
                        const float d2 = xi2 - xr[j * 3 + 2];
                        const float dist = d0 * d0 + d1 * d1 + d2 * d2;
                        if ((double)dist < 0.0001 * 0.0001) { keep = false; break; }
                    }
                    keep_bits |= (keep ? 1u : 0u) << i;
                }
            }
        }
        const int c = __popc(keep_bits);
        // exclusive scan of the counts inside the tile
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wsum[w] = incl;
        __syncthreads();
        int woff = 0, tile_total = 0;
#pragma unroll
        for (int k = 0; k < FCC_ROWS / 64; k++) {
            if (k < w) woff += s_wsum[k];
            tile_total += s_wsum[k];
        }
        const int lo = woff + incl - c;
        s_lo[t] = lo;
        s_kb[t] = (unsigned short)keep_bits;
        // LOCAL: no dependence between tiles -- the tile's candidates are packed at the start of its OWN rows (in place), its total
        // goes to desc[tile] (as int32) for a scan, and ia_deform_pack_tiles moves the blocks to their final place
        if (LOCAL) {
            if (t == 0) { reinterpret_cast<int32_t*>(desc)[tile] = tile_total; s_prefix = (int)(p0 * I); }
        } else
        // decoupled look-back (wave 0): exclusive prefix of the tile over all earlier tiles
        if (w == 0) {
            int excl = 0;
            if (tile == 0) {
                if (lane == 0) __hip_atomic_store(&desc[0], FCC_PREFIX | (uint32_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0)
                    __hip_atomic_store(&desc[tile], FCC_AGGREGATE | (uint32_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // one poll covers FCC_LOOK x 64 predecessors
                int base = tile - 1, part = 0;
                for (;;) {
                    uint64_t d[FCC_LOOK];
#pragma unroll
                    for (int k = 0; k < FCC_LOOK; k++) {
                        const int idx = base - (k * 64 + lane);
                        d[k] = idx >= 0 ? __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FCC_PREFIX;
                    }
                    bool retry = false, done = false;
                    int add = 0;
#pragma unroll
                    for (int k = 0; k < FCC_LOOK; k++) {
                        if (retry || done) continue;                    // wave-uniform
                        const uint32_t status = (uint32_t)(d[k] >> 32);
                        const uint64_t full = __ballot(status == 2u), missing = __ballot(status == 0u);
                        const int first_full = full ? __builtin_ctzll(full) : 64;
                        const uint64_t window = first_full < 63 ? ((1ull << (first_full + 1)) - 1ull) : ~0ull;
                        if (missing & window) { retry = true; continue; }     // a predecessor has not published yet
                        add += ((window >> lane) & 1ull) ? (int)(uint32_t)d[k] : 0;
                        done = first_full < 64;
                    }
                    if (retry) { __builtin_amdgcn_s_sleep(1); continue; }
                    part += add;
                    if (done) break;
                    base -= 64 * FCC_LOOK;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                excl = part;
                if (lane == 0)
                    __hip_atomic_store(&desc[tile], FCC_PREFIX | (uint32_t)(excl + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) s_prefix = excl;
        }
        __syncthreads();
        const int bp = s_prefix;
        if (t < rows) {
            cnt[p0 + t] = c;
            start[p0 + t] = LOCAL ? lo : bp + lo;
        }
        if (!LOCAL && tile == n_tiles - 1 && t == 0) *total = bp + tile_total;
        // item-parallel write-out: consecutive lanes hold consecutive (point, init) items, so the kept ones land on consecutive
        // 12-byte slots of the packed list
        for (int e = t; e < nb; e += FCC_ROWS) {
            const int rr = (int)__umulhi((uint32_t)e, magic_I);
            const int i = e - rr * I;
            const unsigned kb = s_kb[rr];
            const bool kept = (kb >> i) & 1u;
            if (mask) mask[p0 * I + e] = kept ? 1 : 0;
            if (kept) {
                const int64_t dst = (int64_t)bp + s_lo[rr] + __popc(kb & ((1u << i) - 1u));
                cand_x[dst * 3 + 0] = s_x[e * 3 + 0];
                cand_x[dst * 3 + 1] = s_x[e * 3 + 1];
                cand_x[dst * 3 + 2] = s_x[e * 3 + 2];
                if (cand_src) cand_src[dst] = (int32_t)(p0 * I + e);
            }
        }
        __syncthreads();                         // everyone is done with this tile's LDS
    }
}

// second half of the tile-local variant: tile_off = exclusive scan of the tile totals; one workgroup per tile copies the tile's
// packed block from the start of its rows in x to cand_x[tile_off ...] (coalesced on both sides) and turns the tile-local
// starts into global ones.
__global__ __launch_bounds__(FCC_ROWS) void pack_tiles_kernel(int64_t P, int I, const float* __restrict__ x, const int32_t* __restrict__ src_local,
                                                              const int32_t* __restrict__ tile_off, const int32_t* __restrict__ tile_tot,
                                                              int32_t* __restrict__ start, float* __restrict__ cand_x,
                                                              int32_t* __restrict__ cand_src)
{
    const int tile = blockIdx.x, t = threadIdx.x;
    const int64_t p0 = (int64_t)tile * FCC_ROWS;
    const int off = tile_off[tile], n = tile_tot[tile];
    if (p0 + t < P) start[p0 + t] += off;
    const float* in = x + p0 * I * 3;
    float* out = cand_x + (int64_t)off * 3;
    for (int e = t; e < n * 3; e += FCC_ROWS) out[e] = in[e];
    if (cand_src)
        for (int e = t; e < n; e += FCC_ROWS) cand_src[off + e] = src_local[p0 * I + e];
}

// ---- 4'. min over a point's candidates, SDF only (no-grad coarse queries) ----------
__global__ __launch_bounds__(THREADS) void select_min_kernel(int64_t P, const int32_t* __restrict__ start, const int32_t* __restrict__ cnt,
                                                              const float* __restrict__ cand_sdf, const int32_t* __restrict__ order,
                                                              float* __restrict__ sdf_out)
{
    const int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= P) return;
    const int s = start[p], c = cnt[p];
    const int64_t dst = order ? (int64_t)order[p] : p;   // order: the points were evaluated as a permutation of the caller's list
    float best = 1e5f;      // snarf_deformer.py:192
    for (int j = 0; j < c; j++) {
        const float v = cand_sdf[s + j];
        if (v < best) best = v;
    }
    sdf_out[dst] = best;
}

// the same over the SPLIT candidate list of ia_deform_rows_pack_split (first candidates at first_pos[p], the others from n_first on)
__global__ __launch_bounds__(THREADS) void select_min_split_kernel(int64_t P, const int32_t* __restrict__ start, const int32_t* __restrict__ cnt,
                                                                    const int32_t* __restrict__ first_pos, const int32_t* __restrict__ first_tile_off,
                                                                    const int32_t* __restrict__ n_first, const float* __restrict__ cand_sdf,
                                                                    const int32_t* __restrict__ order, float* __restrict__ sdf_out)
{
    const int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= P) return;
    const int c = cnt[p];
    const int64_t dst = order ? (int64_t)order[p] : p;
    float best = 1e5f;      // snarf_deformer.py:192
    if (c > 0) {
        const int64_t fp = (int64_t)first_pos[p] + first_tile_off[p >> 10];         // tiles of 1024 points (ia_deform_rows_pack_split)
        const float v0 = cand_sdf[fp];
        if (v0 < best) best = v0;
        const int64_t tail = (int64_t)*n_first + ((int64_t)start[p] - fp) - 1;
        for (int j = 1; j < c; j++) {
            const float v = cand_sdf[tail + j];
            if (v < best) best = v;
        }
    }
    sdf_out[dst] = best;
}

// ---- 4. select ------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void select_kernel(
    int64_t P, const int32_t* __restrict__ start, const int32_t* __restrict__ cnt, const float* __restrict__ cand_x,
    const int32_t* __restrict__ cand_src, const float* __restrict__ cand_sdf, int sdf_stride,
    const float* __restrict__ cand_grad, const float* __restrict__ cand_feat, int feat_stride, int feat_dim,
    const float* __restrict__ c2w /*[P*I,3,3]*/, float* __restrict__ pts_cano, float* __restrict__ sdf_out,
    uint8_t* __restrict__ valid_out, int32_t* __restrict__ sel_out, float* __restrict__ grad_posed,
    float* __restrict__ grad_cano, float* __restrict__ feat_out)
{
    const int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (p >= P) return;
    const int s = start[p], c = cnt[p];
    float best = 1e5f;      // snarf_deformer.py:192
    int bq = -1;
    for (int j = 0; j < c; j++) {
        const float v = cand_sdf[(int64_t)(s + j) * sdf_stride];
        if (v < best) { best = v; bq = s + j; }
    }
    sdf_out[p] = best;
    valid_out[p] = c > 0 ? 1 : 0;
    if (sel_out) sel_out[p] = bq;
    if (bq >= 0) {
        pts_cano[p * 3 + 0] = cand_x[(int64_t)bq * 3 + 0];
        pts_cano[p * 3 + 1] = cand_x[(int64_t)bq * 3 + 1];
        pts_cano[p * 3 + 2] = cand_x[(int64_t)bq * 3 + 2];
    } else {
        // all candidates invalid (or none below 1e5): torch.min picks slot 0, whose x was zeroed
        // (deformer_torch.py:47-48) unless slot 0 is itself a valid candidate with sdf >= 1e5
        pts_cano[p * 3 + 0] = 0.f; pts_cano[p * 3 + 1] = 0.f; pts_cano[p * 3 + 2] = 0.f;
    }
    if (grad_posed) {
        float g0 = 0.f, g1 = 0.f, g2 = 1.f;       // defaults [0,0,1]: snarf_deformer.py:211-218
        float h0 = 0.f, h1 = 0.f, h2 = 1.f;
        if (bq >= 0) {
            h0 = cand_grad[(int64_t)bq * 3 + 0]; h1 = cand_grad[(int64_t)bq * 3 + 1]; h2 = cand_grad[(int64_t)bq * 3 + 2];
            const float* R = c2w + (int64_t)cand_src[bq] * 9;
            g0 = R[0] * h0 + R[1] * h1 + R[2] * h2;
            g1 = R[3] * h0 + R[4] * h1 + R[5] * h2;
            g2 = R[6] * h0 + R[7] * h1 + R[8] * h2;
        }
        grad_posed[p * 3 + 0] = g0; grad_posed[p * 3 + 1] = g1; grad_posed[p * 3 + 2] = g2;
        if (grad_cano) { grad_cano[p * 3 + 0] = h0; grad_cano[p * 3 + 1] = h1; grad_cano[p * 3 + 2] = h2; }
    }
    if (feat_out)
        for (int k = 0; k < feat_dim; k++)
            feat_out[p * feat_dim + k] = bq >= 0 ? cand_feat[(int64_t)bq * feat_stride + k] : 0.0f;
}

// ---- ray points -----------------------------------------------------------------
// positions = o[ray] + d[ray] * t,  t = t0 (t1 == NULL) or (t0 + t1) / 2
__global__ __launch_bounds__(THREADS) void ray_points_kernel(int64_t n, const float* __restrict__ rays_o,
                                                              const float* __restrict__ rays_d,
                                                              const int64_t* __restrict__ ray_indices,
                                                              const float* __restrict__ t0, const float* __restrict__ t1,
                                                              float* __restrict__ pts)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ray_indices[i];
    const float t = t1 ? (t0[i] + t1[i]) / 2.0f : t0[i];
    pts[i * 3 + 0] = rays_o[r * 3 + 0] + rays_d[r * 3 + 0] * t;
    pts[i * 3 + 1] = rays_o[r * 3 + 1] + rays_d[r * 3 + 1] * t;
    pts[i * 3 + 2] = rays_o[r * 3 + 2] + rays_d[r * 3 + 2] * t;
}

__device__ __forceinline__ void normalize3(float v[3], float eps)
{
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float d = fmaxf(n, eps);
    v[0] /= d; v[1] /= d; v[2] /= d;
}

// shading prep: normal_smpl = normalize(g), normal_world = normalize(g @ R), view_world = normalize(d @ R),
// refl01 = (reflect(-view_world, normal_world) + 1) / 2    (R = w2s[:3,:3]; intrinsic_avatar.py:1059-1061,
// snarf_deformer.py:157-160, radiance.py:123-124, models/utils.py:115-116)
__global__ __launch_bounds__(THREADS) void shade_prep_kernel(int64_t n, const float* __restrict__ sdf_grad,
                                                              const float* __restrict__ rays_d,
                                                              const int64_t* __restrict__ ray_indices,
                                                              const float* __restrict__ R, float* __restrict__ normal_smpl,
                                                              float* __restrict__ normal_world, float* __restrict__ refl01)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ray_indices[i];
    float g[3] = {sdf_grad[i * 3 + 0], sdf_grad[i * 3 + 1], sdf_grad[i * 3 + 2]};
    float d[3] = {rays_d[r * 3 + 0], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    float nw[3], vw[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        nw[c] = g[0] * R[0 * 3 + c] + g[1] * R[1 * 3 + c] + g[2] * R[2 * 3 + c];
        vw[c] = d[0] * R[0 * 3 + c] + d[1] * R[1 * 3 + c] + d[2] * R[2 * 3 + c];
    }
    normalize3(nw, 1e-6f);
    normalize3(vw, 1e-6f);
    normalize3(g, 1e-6f);
    // reflect(x = -vw, n) = 2 dot(x, n) n - x
    const float dt = -(vw[0] * nw[0] + vw[1] * nw[1] + vw[2] * nw[2]);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        normal_smpl[i * 3 + c] = g[c];
        normal_world[i * 3 + c] = nw[c];
        refl01[i * 3 + c] = ((2.0f * dt * nw[c] + vw[c]) + 1.0f) / 2.0f;
    }
}

// backward of shade_prep w.r.t. sdf_grad: g_nw (direct) and g_refl01 (through the SH input) -> g_sdf_grad
__global__ __launch_bounds__(THREADS) void shade_prep_bwd_kernel(int64_t n, const float* __restrict__ sdf_grad,
                                                                  const float* __restrict__ rays_d,
                                                                  const int64_t* __restrict__ ray_indices,
                                                                  const float* __restrict__ R, const float* __restrict__ g_nw,
                                                                  const float* __restrict__ g_refl01, const float* __restrict__ g_ns,
                                                                  float* __restrict__ g_sdf_grad)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ray_indices[i];
    const float g[3] = {sdf_grad[i * 3 + 0], sdf_grad[i * 3 + 1], sdf_grad[i * 3 + 2]};
    const float d[3] = {rays_d[r * 3 + 0], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    float nwu[3], vw[3], nw[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        nwu[c] = g[0] * R[0 * 3 + c] + g[1] * R[1 * 3 + c] + g[2] * R[2 * 3 + c];
        vw[c] = d[0] * R[0 * 3 + c] + d[1] * R[1 * 3 + c] + d[2] * R[2 * 3 + c];
    }
    normalize3(vw, 1e-6f);
    const float len = sqrtf(nwu[0] * nwu[0] + nwu[1] * nwu[1] + nwu[2] * nwu[2]);
    const float den = fmaxf(len, 1e-6f);
#pragma unroll
    for (int c = 0; c < 3; c++) nw[c] = nwu[c] / den;
    const float dt = -(vw[0] * nw[0] + vw[1] * nw[1] + vw[2] * nw[2]);
    float G[3];
    float gr_dot_nw = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) gr_dot_nw += 0.5f * (g_refl01 ? g_refl01[i * 3 + c] : 0.f) * nw[c];
#pragma unroll
    for (int c = 0; c < 3; c++)
        G[c] = (g_nw ? g_nw[i * 3 + c] : 0.f) + 2.0f * dt * 0.5f * (g_refl01 ? g_refl01[i * 3 + c] : 0.f) - 2.0f * gr_dot_nw * vw[c];
    float Gu[3];
    if (len > 1e-6f) {
        const float dn = nw[0] * G[0] + nw[1] * G[1] + nw[2] * G[2];
#pragma unroll
        for (int c = 0; c < 3; c++) Gu[c] = (G[c] - nw[c] * dn) / len;
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) Gu[c] = G[c] / 1e-6f;
    }
    // normal_smpl = g / max(|g|, 1e-6) (the BRDF's normal in the PBR branch): its gradient joins here instead of in five torch launches
    float gs[3] = {0.f, 0.f, 0.f};
    if (g_ns) {
        const float nrm = fmaxf(sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), 1e-6f);
        const float ns[3] = {g[0] / nrm, g[1] / nrm, g[2] / nrm};
        const float dsn = g_ns[i * 3 + 0] * ns[0] + g_ns[i * 3 + 1] * ns[1] + g_ns[i * 3 + 2] * ns[2];
#pragma unroll
        for (int c = 0; c < 3; c++) gs[c] = (g_ns[i * 3 + c] - dsn * ns[c]) / nrm;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) g_sdf_grad[i * 3 + a] = Gu[0] * R[a * 3 + 0] + Gu[1] * R[a * 3 + 1] + Gu[2] * R[a * 3 + 2] + gs[a];
}

// alpha = 1 - exp(-sigma(sdf) * dist), sigma = Laplace CDF density (density.py:25-30)
__global__ __launch_bounds__(THREADS) void laplace_alpha_kernel(int64_t n, const float* __restrict__ sdf,
                                                                 const float* __restrict__ dists, float dist_const,
                                                                 const float* __restrict__ beta_p, float* __restrict__ alpha)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float beta = *beta_p;
    const float s = sdf[i];
    const float sg = (float)((s > 0.f) - (s < 0.f));
    const float dens = (1.0f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
    const float dist = dists ? dists[i] : dist_const;
    alpha[i] = 1.0f - expf(-dens * dist);
}

// the same with the interval length taken from the interval's ends (dist = t_end - t_start: the subtraction the caller would otherwise
// materialise -- 303 M elements per headline step)
__global__ __launch_bounds__(THREADS) void laplace_alpha_intervals_kernel(int64_t n, const float* __restrict__ sdf, const float* __restrict__ t_starts,
                                                                           const float* __restrict__ t_ends, const float* __restrict__ beta_p,
                                                                           float* __restrict__ alpha)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float beta = *beta_p;
    const float s = sdf[i];
    const float sg = (float)((s > 0.f) - (s < 0.f));
    const float dens = (1.0f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
    const float dist = t_ends[i] - t_starts[i];
    alpha[i] = 1.0f - expf(-dens * dist);
}

// backward: g_sdf[i] = g_alpha * d alpha / d sdf ; g_beta partial sums -> atomicAdd
__global__ __launch_bounds__(THREADS) void laplace_alpha_bwd_kernel(int64_t n, const float* __restrict__ sdf,
                                                                     const float* __restrict__ dists, float dist_const,
                                                                     const float* __restrict__ beta_p,
                                                                     const float* __restrict__ g_alpha,
                                                                     float* __restrict__ g_sdf, float* __restrict__ g_beta)
{
    // grid-stride: a workgroup accumulates its share of d L / d beta in registers and issues ONE atomic at the end
    // (one atomic per wave on a single address serialised at ~12 ns each: 69 k of them = 0.8 ms for 4.4 M samples)
    __shared__ float s_part[THREADS / 64];
    float gb = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float beta = *beta_p;
        const float s = sdf[i], as = fabsf(s);
        const float sg = (float)((s > 0.f) - (s < 0.f));
        const float e = expf(-as / beta);                 // expm1 + 1
        const float dens = (1.0f / beta) * (0.5f + 0.5f * sg * (e - 1.0f));
        const float dist = dists ? dists[i] : dist_const;
        const float ga = g_alpha[i] * dist * expf(-dens * dist);      // d alpha / d dens
        // d dens / d s = (1/beta) * 0.5 * sg * e * (-sg/beta) = -e / (2 beta^2)   (s != 0)
        g_sdf[i] = ga * (-(e) / (2.0f * beta * beta)) * (s == 0.f ? 0.f : 1.f);
        // d dens / d beta = -dens/beta + (1/beta) * 0.5 * sg * e * (as / beta^2)
        gb += ga * (-dens / beta + 0.5f * sg * e * as / (beta * beta * beta));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gb += __shfl_down(gb, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = gb;
    __syncthreads();
    if (threadIdx.x == 0 && g_beta) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < THREADS / 64; w++) t += s_part[w];
        if (t != 0.0f) atomicAdd(g_beta, t);
    }
}

}  // namespace

IA_EXPORT int ia_deform_filter_count(int64_t P, int I, const float* x, const uint8_t* valid, uint8_t* mask,
                                     int32_t* cnt, ia_stream_t stream)
{
    if (P == 0) return IA_OK;
    IA_REQUIRE(I >= 1 && I <= FC_MAX_I, "ia_deform_filter_count: at most 16 initialisations per point");
    const size_t lds = (size_t)THREADS * I * 3 * sizeof(float) + (size_t)THREADS * I + 16;
    filter_count_kernel<<<ia::cdiv(P, THREADS), THREADS, lds, (hipStream_t)stream>>>(P, I, x, valid, mask, cnt);
    return ia::check_launch("ia_deform_filter_count");
}

IA_EXPORT int ia_deform_compact(int64_t P, int I, const float* x, const uint8_t* mask, const int32_t* start,
                                float* cand_x, int32_t* cand_src, ia_stream_t stream)
{
    if (P == 0) return IA_OK;
    compact_fill_kernel<<<ia::cdiv(P, THREADS), THREADS, 0, (hipStream_t)stream>>>(P, I, x, mask, start, cand_x, cand_src);
    return ia::check_launch("ia_deform_compact");
}

IA_EXPORT size_t ia_deform_filter_compact_tmp_bytes(int64_t P)
{
    return (size_t)(ia::cdiv(P > 0 ? P : 1, FCC_ROWS)) * sizeof(uint64_t) + 16;
}

IA_EXPORT int ia_deform_filter_compact(int64_t P, int I, const float* x, const uint8_t* valid, int32_t* cnt, int32_t* start,
                                       float* cand_x, int32_t* cand_src, uint8_t* mask, int32_t* total, void* tmp, size_t tmp_bytes,
                                       ia_stream_t stream)
{
    IA_REQUIRE(total != nullptr, "ia_deform_filter_compact: total is required");
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) {
        if (hipMemsetAsync(total, 0, sizeof(int32_t), s) != hipSuccess) return ia::check_launch("ia_deform_filter_compact(memset)");
        return IA_OK;
    }
    IA_REQUIRE(I >= 1 && I <= FC_MAX_I, "ia_deform_filter_compact: at most 16 initialisations per point");
    IA_REQUIRE(P * I < ((int64_t)1 << 31), "ia_deform_filter_compact: P * I must stay below 2^31");
    const int64_t n_tiles = ia::cdiv(P, FCC_ROWS);
    const size_t need = ia_deform_filter_compact_tmp_bytes(P);
    IA_REQUIRE(tmp != nullptr && tmp_bytes >= need, "ia_deform_filter_compact: tmp too small (ia_deform_filter_compact_tmp_bytes)");
    IA_REQUIRE((reinterpret_cast<uintptr_t>(tmp) & 7) == 0, "ia_deform_filter_compact: tmp must be 8-byte aligned");
    if (hipMemsetAsync(tmp, 0, need, s) != hipSuccess) return ia::check_launch("ia_deform_filter_compact(memset)");
    uint64_t* desc = reinterpret_cast<uint64_t*>(tmp);
    const uint32_t magic = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)I - 1) / (uint64_t)I);   // e / I for e < 2^16 (checked for I <= 16)
    const size_t lds = (size_t)FCC_ROWS * I * 3 * sizeof(float) + (size_t)FCC_ROWS * I + 16;
    const unsigned grid = (unsigned)n_tiles;
    if (I == 13)      // the reference's 13 initialisations (snarf_deformer.py:98): straight-line filter
        filter_compact_kernel<13, false><<<grid, FCC_ROWS, lds, s>>>(P, I, magic, (int)n_tiles, x, valid, desc, cnt, start, cand_x, cand_src,
                                                                      mask, total);
    else
        filter_compact_kernel<0, false><<<grid, FCC_ROWS, lds, s>>>(P, I, magic, (int)n_tiles, x, valid, desc, cnt, start, cand_x, cand_src,
                                                                     mask, total);
    return ia::check_launch("ia_deform_filter_compact");
}

// The same result in two steps without any dependence between tiles (the look-back above runs at half the rate of its loads):
//   ia_deform_filter_tiles : filter + count; every tile's candidates packed IN PLACE at the start of the tile's own rows of x
//                            (x is consumed), cand_src likewise into src_local [P*I] (optional); cnt [P], tile-local start [P];
//                            tile totals -> exclusive scan -> tile_off [n_tiles], *total = Q
//   (host reads Q, allocates cand_x [Q,3])
//   ia_deform_pack_tiles   : segmented copy of the tile blocks to cand_x (must NOT alias x), start += tile_off
// tmp: ia_deform_filter_tiles_tmp_bytes(P) bytes, 8-byte aligned, kept between the two calls.
IA_EXPORT size_t ia_deform_filter_tiles_tmp_bytes(int64_t P)
{
    const int64_t n_tiles = ia::cdiv(P > 0 ? P : 1, FCC_ROWS);
    return (size_t)n_tiles * 8 + (size_t)ia_scan_tmp_bytes(n_tiles) + 64;
}

IA_EXPORT int ia_deform_filter_tiles(int64_t P, int I, float* x, const uint8_t* valid, int32_t* cnt, int32_t* start, int32_t* src_local,
                                     uint8_t* mask, int32_t* total, void* tmp, size_t tmp_bytes, ia_stream_t stream)
{
    IA_REQUIRE(total != nullptr, "ia_deform_filter_tiles: total is required");
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) {
        if (hipMemsetAsync(total, 0, sizeof(int32_t), s) != hipSuccess) return ia::check_launch("ia_deform_filter_tiles(memset)");
        return IA_OK;
    }
    IA_REQUIRE(I >= 1 && I <= FC_MAX_I, "ia_deform_filter_tiles: at most 16 initialisations per point");
    IA_REQUIRE(P * I < ((int64_t)1 << 31), "ia_deform_filter_tiles: P * I must stay below 2^31");
    const int64_t n_tiles = ia::cdiv(P, FCC_ROWS);
    IA_REQUIRE(tmp != nullptr && tmp_bytes >= ia_deform_filter_tiles_tmp_bytes(P), "ia_deform_filter_tiles: tmp too small");
    IA_REQUIRE((reinterpret_cast<uintptr_t>(tmp) & 7) == 0, "ia_deform_filter_tiles: tmp must be 8-byte aligned");
    int32_t* tile_tot = reinterpret_cast<int32_t*>(tmp);
    int32_t* tile_off = tile_tot + n_tiles;
    void* scan_tmp = reinterpret_cast<char*>(tmp) + (size_t)n_tiles * 8;
    const uint32_t magic = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)I - 1) / (uint64_t)I);
    const size_t lds = (size_t)FCC_ROWS * I * 3 * sizeof(float) + (size_t)FCC_ROWS * I + 16;
    const unsigned grid = (unsigned)n_tiles;
    if (I == 13)
        filter_compact_kernel<13, true><<<grid, FCC_ROWS, lds, s>>>(P, I, magic, (int)n_tiles, x, valid, reinterpret_cast<uint64_t*>(tile_tot),
                                                                     cnt, start, x, src_local, mask, total);
    else
        filter_compact_kernel<0, true><<<grid, FCC_ROWS, lds, s>>>(P, I, magic, (int)n_tiles, x, valid, reinterpret_cast<uint64_t*>(tile_tot),
                                                                    cnt, start, x, src_local, mask, total);
    int r = ia::check_launch("ia_deform_filter_tiles");
    if (r != IA_OK) return r;
    return ia_exclusive_scan_i32(tile_tot, tile_off, total, n_tiles, scan_tmp, stream);
}

IA_EXPORT int ia_deform_pack_tiles(int64_t P, int I, const float* x, const int32_t* src_local, int32_t* start, float* cand_x,
                                   int32_t* cand_src, const void* tmp, ia_stream_t stream)
{
    if (P == 0) return IA_OK;
    IA_REQUIRE(cand_x != x, "ia_deform_pack_tiles: cand_x must not alias x");
    IA_REQUIRE((cand_src == nullptr) == (src_local == nullptr), "ia_deform_pack_tiles: cand_src and src_local come together");
    const int64_t n_tiles = ia::cdiv(P, FCC_ROWS);
    const int32_t* tile_tot = reinterpret_cast<const int32_t*>(tmp);
    const int32_t* tile_off = tile_tot + n_tiles;
    pack_tiles_kernel<<<(unsigned)n_tiles, FCC_ROWS, 0, (hipStream_t)stream>>>(P, I, x, src_local, tile_off, tile_tot, start, cand_x,
                                                                                cand_src);
    return ia::check_launch("ia_deform_pack_tiles");
}

IA_EXPORT int ia_deform_select(int64_t P, const int32_t* start, const int32_t* cnt, const float* cand_x,
                               const int32_t* cand_src, const float* cand_sdf, int sdf_stride, const float* cand_grad,
                               const float* cand_feat, int feat_stride, int feat_dim, const float* c2w,
                               float* pts_cano, float* sdf, uint8_t* valid, int32_t* sel, float* grad_posed,
                               float* grad_cano, float* feat, ia_stream_t stream)
{
    if (P == 0) return IA_OK;
    IA_REQUIRE((grad_posed == nullptr) || (cand_grad != nullptr && c2w != nullptr), "grad_posed needs cand_grad and c2w");
    select_kernel<<<ia::cdiv(P, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        P, start, cnt, cand_x, cand_src, cand_sdf, sdf_stride, cand_grad, cand_feat, feat_stride, feat_dim, c2w,
        pts_cano, sdf, valid, sel, grad_posed, grad_cano, feat);
    return ia::check_launch("ia_deform_select");
}

IA_EXPORT int ia_deform_select_min(int64_t P, const int32_t* start, const int32_t* cnt, const float* cand_sdf, float* sdf,
                                   ia_stream_t stream)
{
    if (P == 0) return IA_OK;
    select_min_kernel<<<ia::cdiv(P, THREADS), THREADS, 0, (hipStream_t)stream>>>(P, start, cnt, cand_sdf, nullptr, sdf);
    return ia::check_launch("ia_deform_select_min");
}

// the same, for points that were evaluated in another order than the caller's: sdf[order[p]] = min over the candidates of p
IA_EXPORT int ia_deform_select_min_split(int64_t P, const int32_t* start, const int32_t* cnt, const int32_t* first_pos, const int32_t* first_tile_off,
                                         const int32_t* n_first, const float* cand_sdf, const int32_t* order, float* sdf, ia_stream_t stream)
{
    if (P == 0) return IA_OK;
    IA_REQUIRE(first_pos != nullptr && first_tile_off != nullptr && n_first != nullptr, "ia_deform_select_min_split: first_pos, first_tile_off and n_first are required");
    select_min_split_kernel<<<ia::cdiv(P, THREADS), THREADS, 0, (hipStream_t)stream>>>(P, start, cnt, first_pos, first_tile_off, n_first, cand_sdf, order, sdf);
    return ia::check_launch("ia_deform_select_min_split");
}

IA_EXPORT int ia_deform_select_min_scatter(int64_t P, const int32_t* start, const int32_t* cnt, const float* cand_sdf,
                                           const int32_t* order, float* sdf, ia_stream_t stream)
{
    if (P == 0) return IA_OK;
    IA_REQUIRE(order != nullptr, "ia_deform_select_min_scatter: order is required");
    select_min_kernel<<<ia::cdiv(P, THREADS), THREADS, 0, (hipStream_t)stream>>>(P, start, cnt, cand_sdf, order, sdf);
    return ia::check_launch("ia_deform_select_min_scatter");
}

IA_EXPORT int ia_ray_points(int64_t n, const float* rays_o, const float* rays_d, const int64_t* ray_indices,
                            const float* t0, const float* t1, float* pts, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    ray_points_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, rays_o, rays_d, ray_indices, t0, t1, pts);
    return ia::check_launch("ia_ray_points");
}

IA_EXPORT int ia_shade_prep(int64_t n, const float* sdf_grad, const float* rays_d, const int64_t* ray_indices,
                            const float* w2s_rot, float* normal_smpl, float* normal_world, float* refl01,
                            ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    shade_prep_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, sdf_grad, rays_d, ray_indices, w2s_rot,
                                                                               normal_smpl, normal_world, refl01);
    return ia::check_launch("ia_shade_prep");
}

IA_EXPORT int ia_shade_prep_bwd(int64_t n, const float* sdf_grad, const float* rays_d, const int64_t* ray_indices,
                                const float* w2s_rot, const float* g_normal_world, const float* g_refl01,
                                const float* g_normal_smpl, float* g_sdf_grad, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    shade_prep_bwd_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, sdf_grad, rays_d, ray_indices, w2s_rot,
                                                                                   g_normal_world, g_refl01, g_normal_smpl, g_sdf_grad);
    return ia::check_launch("ia_shade_prep_bwd");
}

IA_EXPORT int ia_laplace_alpha(int64_t n, const float* sdf, const float* dists, float dist_const, const float* beta,
                               float* alpha, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    laplace_alpha_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, sdf, dists, dist_const, beta, alpha);
    return ia::check_launch("ia_laplace_alpha");
}

IA_EXPORT int ia_laplace_alpha_intervals(int64_t n, const float* sdf, const float* t_starts, const float* t_ends, const float* beta,
                                         float* alpha, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    laplace_alpha_intervals_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, sdf, t_starts, t_ends, beta, alpha);
    return ia::check_launch("ia_laplace_alpha_intervals");
}

IA_EXPORT int ia_laplace_alpha_bwd(int64_t n, const float* sdf, const float* dists, float dist_const,
                                   const float* beta, const float* g_alpha, float* g_sdf, float* g_beta,
                                   ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    int grid = ia::cdiv(n, THREADS);
    if (grid > 2048) grid = 2048;
    laplace_alpha_bwd_kernel<<<grid, THREADS, 0, (hipStream_t)stream>>>(n, sdf, dists, dist_const, beta, g_alpha, g_sdf, g_beta);
    return ia::check_launch("ia_laplace_alpha_bwd");
}

// ---- winners -> shading inputs, and the eikonal term: the elementwise glue of the differentiable shading pass --------
// SNARFDeformer.deform's tail (snarf_deformer.py:192-231) as ONE kernel per direction instead of ~20 elementwise torch
// launches over 4.4 M rows: features masked by `valid`, sdf default 1e5, normal push-forward c2w . grad_c with the
// default (0, 0, 1) for points without a canonical correspondence; c2w = fwd_J of the winning candidate.
namespace {

__global__ __launch_bounds__(THREADS) void select_push_kernel(int64_t n, const float* __restrict__ out13,
                                                               const float* __restrict__ grad_c,
                                                               const uint8_t* __restrict__ valid,
                                                               const float* __restrict__ fwd_J,
                                                               const int32_t* __restrict__ cand_src,
                                                               const int32_t* __restrict__ sel, float* __restrict__ feat,
                                                               float* __restrict__ sdf, float* __restrict__ sdf_grad,
                                                               float* __restrict__ c2w)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const bool v = valid[i] != 0;
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = 0.0f;
    if (fwd_J) {                                   // same gather as fwd_J[cand_src[clamp(sel, 0)]] (also for invalid points)
        int s = sel[i];
        if (s < 0) s = 0;
        const float* src = fwd_J + (int64_t)cand_src[s] * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = src[k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) c2w[i * 9 + k] = R[k];
#pragma unroll
    for (int k = 0; k < 13; k++) feat[i * 13 + k] = v ? out13[i * 13 + k] : 0.0f;
    sdf[i] = v ? out13[i * 13] : 1e5f;
    const float g0 = grad_c[i * 3 + 0], g1 = grad_c[i * 3 + 1], g2 = grad_c[i * 3 + 2];
    sdf_grad[i * 3 + 0] = v ? R[0] * g0 + R[1] * g1 + R[2] * g2 : 0.0f;
    sdf_grad[i * 3 + 1] = v ? R[3] * g0 + R[4] * g1 + R[5] * g2 : 0.0f;
    sdf_grad[i * 3 + 2] = v ? R[6] * g0 + R[7] * g1 + R[8] * g2 : 1.0f;
}

__global__ __launch_bounds__(THREADS) void select_push_bwd_kernel(int64_t n, const uint8_t* __restrict__ valid,
                                                                   const float* __restrict__ c2w,
                                                                   const float* __restrict__ g_feat,
                                                                   const float* __restrict__ g_sdf,
                                                                   const float* __restrict__ g_sdf_grad,
                                                                   float* __restrict__ g_out13, float* __restrict__ g_grad_c)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const bool v = valid[i] != 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        float g = (v && g_feat) ? g_feat[i * 13 + k] : 0.0f;
        if (k == 0 && v && g_sdf) g += g_sdf[i];
        g_out13[i * 13 + k] = g;
    }
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    if (v && g_sdf_grad) {
        const float* R = c2w + i * 9;
        const float a = g_sdf_grad[i * 3 + 0], b = g_sdf_grad[i * 3 + 1], c = g_sdf_grad[i * 3 + 2];
        o0 = R[0] * a + R[3] * b + R[6] * c;
        o1 = R[1] * a + R[4] * b + R[7] * c;
        o2 = R[2] * a + R[5] * b + R[8] * c;
    }
    g_grad_c[i * 3 + 0] = o0; g_grad_c[i * 3 + 1] = o1; g_grad_c[i * 3 + 2] = o2;
}

// eikonal term over the valid samples: per-workgroup partial sums of (|g| - 1)^2 and of the valid count (the host-side
// reduction over the few thousand partials is a deterministic torch sum: no float atomics)
constexpr int EIK_PER_WG = 4 * THREADS;
__global__ __launch_bounds__(THREADS) void eikonal_kernel(int64_t n, const float* __restrict__ g, const uint8_t* __restrict__ valid,
                                                           float* __restrict__ partial /*[nwg,2]*/)
{
    __shared__ float s_sum[THREADS / 64], s_cnt[THREADS / 64];
    float s = 0.0f, c = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int64_t i = (int64_t)blockIdx.x * EIK_PER_WG + r * THREADS + threadIdx.x;
        if (i < n && valid[i]) {
            const float x = g[i * 3 + 0], y = g[i * 3 + 1], z = g[i * 3 + 2];
            const float d = sqrtf(x * x + y * y + z * z) - 1.0f;
            s += d * d;
            c += 1.0f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); c += __shfl_xor(c, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = s; s_cnt[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.0f, tc = 0.0f;
#pragma unroll
        for (int w = 0; w < THREADS / 64; w++) { ts += s_sum[w]; tc += s_cnt[w]; }
        partial[blockIdx.x * 2 + 0] = ts;
        partial[blockIdx.x * 2 + 1] = tc;
    }
}

// d/dg of  w * sum_valid (|g| - 1)^2 :  w * 2 (|g| - 1) g / |g|   (0 where |g| = 0, like torch's norm backward)
__global__ __launch_bounds__(THREADS) void eikonal_bwd_kernel(int64_t n, const float* __restrict__ g, const uint8_t* __restrict__ valid,
                                                               const float* __restrict__ w, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
    if (valid[i]) {
        const float x = g[i * 3 + 0], y = g[i * 3 + 1], z = g[i * 3 + 2];
        const float nrm = sqrtf(x * x + y * y + z * z);
        if (nrm > 0.0f) {
            const float k = w[0] * 2.0f * (nrm - 1.0f) / nrm;
            o0 = k * x; o1 = k * y; o2 = k * z;
        }
    }
    out[i * 3 + 0] = o0; out[i * 3 + 1] = o1; out[i * 3 + 2] = o2;
}

}  // namespace

IA_EXPORT int ia_select_push(int64_t n, const float* out13, const float* grad_c, const uint8_t* valid, const float* fwd_J,
                             const int32_t* cand_src, const int32_t* sel, float* feat, float* sdf, float* sdf_grad,
                             float* c2w, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(out13 && grad_c && valid && feat && sdf && sdf_grad && c2w, "null buffer");
    IA_REQUIRE(fwd_J == nullptr || (cand_src && sel), "fwd_J needs cand_src and sel");
    select_push_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, out13, grad_c, valid, fwd_J, cand_src, sel,
                                                                                   feat, sdf, sdf_grad, c2w);
    return ia::check_launch("ia_select_push");
}

IA_EXPORT int ia_select_push_bwd(int64_t n, const uint8_t* valid, const float* c2w, const float* g_feat, const float* g_sdf,
                                 const float* g_sdf_grad, float* g_out13, float* g_grad_c, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(valid && c2w && g_out13 && g_grad_c, "null buffer");
    select_push_bwd_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, valid, c2w, g_feat, g_sdf, g_sdf_grad,
                                                                                       g_out13, g_grad_c);
    return ia::check_launch("ia_select_push_bwd");
}

IA_EXPORT int64_t ia_eikonal_partials(int64_t n) { return n <= 0 ? 0 : (n + EIK_PER_WG - 1) / EIK_PER_WG; }

IA_EXPORT int ia_eikonal(int64_t n, const float* sdf_grad, const uint8_t* valid, float* partial, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(sdf_grad && valid && partial, "null buffer");
    eikonal_kernel<<<(int)ia_eikonal_partials(n), THREADS, 0, (hipStream_t)stream>>>(n, sdf_grad, valid, partial);
    return ia::check_launch("ia_eikonal");
}

IA_EXPORT int ia_eikonal_bwd(int64_t n, const float* sdf_grad, const uint8_t* valid, const float* weight, float* g_sdf_grad,
                             ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(sdf_grad && valid && weight && g_sdf_grad, "null buffer");
    eikonal_bwd_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, sdf_grad, valid, weight, g_sdf_grad);
    return ia::check_launch("ia_eikonal_bwd");
}
