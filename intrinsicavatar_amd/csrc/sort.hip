// sort.hip -- spatial ordering of large query batches (render.py:_sdf_at): Morton keys of the points' cells, a hand-written
// stable LSD radix sort of (key, int32 index) pairs over the key bits that carry locality, and the index-driven row gather /
// scatter.  The order only schedules the work: results are written back through the same indices.
//
// The sort (round 5; rounds 2-4 called rocPRIM's onesweep here).  Onesweep's decoupled look-back makes every workgroup of a pass
// wait for its predecessors' tile descriptors; with the step's second HIP stream holding the CUs those waits become spinning
// (profiles/r04_headline_kernel_stats_two_streams.csv: 1 385 instead of 272 us per launch).  Here no workgroup ever waits for
// another one -- per digit pass three plain kernels:
//   count    one workgroup per tile of TILE consecutive elements: LDS histogram of the tile's digits -> hist[tile][digit]
//            (pass 0 also computes the Morton keys from the points and stores them: the points are read once)
//   scan     exclusive prefix sum of the [tile][digit] count matrix in digit-major order (chunk sums -> bases -> apply, no look-back)
//   scatter  the same tile again: every element's STABLE rank among the tile's elements of its digit, by wave ballots --
//            the peers of a lane (same digit, same wave instruction) come from `bits` ballots, the wave keeps running per-digit
//            counts of its own contiguous segment in LDS, and the waves' segments are stitched by a prefix over the waves --;
//            the (key, index) pair goes to LDS at digit_start + rank, and the tile is written out in that order, so consecutive lanes
//            store consecutive elements of a digit's run (coalesced runs of TILE / 2^bits elements of 8 bytes on average).
// Element order inside a tile is (wave, round, lane) = input order, every pass is stable, so the result is the stable argsort of
// the key bits [drop_bits, 30).  Traffic per element with three 10-bit passes: 12 + 4 (keys) + 4 + 8 in pass 0, 8 + 8 + 8 in pass 1,
// 8 + 8 + 4 in the last one: 72 B.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "ia_common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    return (v | (v << 2)) & 0x09249249u;
}

__global__ __launch_bounds__(THREADS) void gather_rows3_i32_kernel(int64_t n, const float* __restrict__ src, const int32_t* __restrict__ order,
                                                                    float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t j = order[i];
    dst[3 * i] = src[3 * j]; dst[3 * i + 1] = src[3 * j + 1]; dst[3 * i + 2] = src[3 * j + 2];
}

__global__ __launch_bounds__(THREADS) void scatter_f32_i32_kernel(int64_t n, const float* __restrict__ src, const int32_t* __restrict__ order,
                                                                   float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    dst[order[i]] = src[i];
}

__global__ __launch_bounds__(THREADS) void scatter_rows3_i32_kernel(int64_t n, const float* __restrict__ src, const int32_t* __restrict__ order,
                                                                     float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t j = order[i];
    dst[3 * j] = src[3 * i]; dst[3 * j + 1] = src[3 * i + 1]; dst[3 * j + 2] = src[3 * i + 2];
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- the radix sort -------------------------------------------------------------------------------------------------------------
// measured on the MI355X (tools/sort_probe.py, 18 M / 40 M / 140 M march points, ms per sort; -DIA_SORT_THREADS / -DIA_SORT_BITS builds):
//   rocPRIM onesweep (rounds 2-4)                  0.732 / 1.394 / 4.449
//   256 threads, 10-bit digits, separate columns   0.585 / 1.099 / 4.197     (8-bit digits, four passes: 0.566 / 1.099 / 3.389)
//   256 threads, 10-bit digits, pair column        0.536 / 1.040 / 3.481
//   512 threads, 10-bit digits, pair column        0.504 / 0.984 / 3.018     <- default (8-bit: 0.565 / 1.121 / 3.486)
// what a pass costs is the length of the runs it writes (TILE / 2^bits elements): one 16 384-element tile per CU beats two of 8192.
#ifndef IA_SORT_THREADS
#define IA_SORT_THREADS 512
#endif
constexpr int ST = IA_SORT_THREADS;      // threads per workgroup
constexpr int SW = ST / 64;              // waves
constexpr int SI = 32;                   // elements per thread
constexpr int TILE = ST * SI;            // 16 384 elements per workgroup
constexpr int SEG = 64 * SI;             // a wave's contiguous segment of the tile
#ifndef IA_SORT_BITS
#define IA_SORT_BITS 10
#endif
constexpr int MAXB = IA_SORT_BITS;       // widest digit
constexpr int NBMAX = 1 << MAXB;
constexpr int CT = 256;                  // threads of the count kernel (same tiles)
constexpr int CI = TILE / CT;

// raster > 0 (experiment, IA_SORT_RASTER_BITS): the low 3 * raster key bits order the cells of a 2^raster-cube x-fastest (z, y, x)
// instead of bit-interleaved -- consecutive points then run along x, the direction in which tiny-cuda-nn's hash keeps 16 neighbouring
// cells in one 128-byte line.  Any bijection of the cell bits is a valid schedule (results do not depend on the order).
__device__ __forceinline__ uint32_t morton_key(const float* __restrict__ pts, int64_t i, float ox, float oy, float oz, float inv_cell, int raster = 0)
{
    const float fx = (pts[3 * i] - ox) * inv_cell, fy = (pts[3 * i + 1] - oy) * inv_cell, fz = (pts[3 * i + 2] - oz) * inv_cell;
    const uint32_t x = (uint32_t)fminf(fmaxf(fx, 0.0f), 1023.0f), y = (uint32_t)fminf(fmaxf(fy, 0.0f), 1023.0f),
                   z = (uint32_t)fminf(fmaxf(fz, 0.0f), 1023.0f);
    if (raster > 0) {
        const uint32_t m = (1u << raster) - 1u;
        const uint32_t hi = spread10(x >> raster) | (spread10(y >> raster) << 1) | (spread10(z >> raster) << 2);
        return (hi << (3 * raster)) | ((z & m) << (2 * raster)) | ((y & m) << raster) | (x & m);
    }
    return spread10(x) | (spread10(y) << 1) | (spread10(z) << 2);
}

// peers of this lane: the active lanes of the wave instruction whose digit equals this lane's (`bits` ballots; ~7 VALU instructions
// per bit: the fallback ranking, see wave_rank)
__device__ __forceinline__ uint64_t digit_peers(uint32_t digit, int bits, bool valid)
{
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < bits; b++) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// hist[tile * nb + digit] = number of elements of the tile with that digit (one contiguous row per tile: a digit-major matrix made every
// workgroup write -- and the scatter read -- 1024 isolated 4-byte words, a sector each).  FIRST: keys are computed from the points and stored.
// (Counting needs no order: one LDS atomic per lane -- up to 64 lanes on one address when a tile's high digits are constant, which the
//  LDS serialises in 64 cycles; finding a lane's peers by ballots first, to issue one atomic per distinct digit, costs 10 x 7 VALU
//  instructions x 4 cycles: measured 1.4 TB/s of key reads.  A wave whose lanes all hold one digit adds its count in one atomic.)
template <bool FIRST>
__global__ __launch_bounds__(CT) void radix_count_kernel(int64_t n, const float* __restrict__ pts, float ox, float oy, float oz, float inv_cell,
                                                          uint32_t* __restrict__ keys /* FIRST: out */, const uint2* __restrict__ pairs /* !FIRST: in */,
                                                          int shift, int bits, int32_t* __restrict__ hist, int ntiles, int raster)
{
    __shared__ int32_t s_hist[NBMAX];
    const int nb = 1 << bits;
    const uint32_t mask = (uint32_t)nb - 1u;
    for (int d = threadIdx.x; d < nb; d += CT) s_hist[d] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.x * TILE + threadIdx.x;
    for (int r0 = 0; r0 < CI; r0 += 8) {
        uint32_t key[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int64_t i = base + (int64_t)(r0 + r) * CT;
            key[r] = 0;
            if (i < n) {
                if (FIRST) { key[r] = morton_key(pts, i, ox, oy, oz, inv_cell, raster); keys[i] = key[r]; }
                else key[r] = pairs[i].x;
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool valid = base + (int64_t)(r0 + r) * CT < n;
            const uint32_t digit = (key[r] >> shift) & mask;
            const uint64_t vm = __ballot(valid);
            if (vm == 0ull) continue;
            const uint32_t d0 = __builtin_amdgcn_readlane(digit, __ffsll((long long)vm) - 1);
            if (__ballot(valid && digit != d0) == 0ull) {
                if (lane == __ffsll((long long)vm) - 1) atomicAdd(&s_hist[d0], __popcll(vm));
            } else if (valid) {
                atomicAdd(&s_hist[digit], 1);
            }
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += CT) hist[(int64_t)blockIdx.x * nb + d] = s_hist[d];
}

// ---- the scan of the [tile][digit] count matrix in digit-major order: offs[t][d] = sum_{d' < d} total[d'] + sum_{t' < t} hist[t'][d] ----
// three plain kernels over chunks of SCAN_CHUNK tiles (rows are read and written whole: coalesced), nothing waits for anything:
//   sums    chunk_sum[c][d] = sum of the chunk's rows
//   bases   one workgroup: per digit the exclusive prefix over the chunks, then the exclusive scan of the digit totals -> chunk_off[c][d]
//   apply   offs[t][d] = chunk_off[c][d] + the chunk's rows above t
constexpr int SCAN_CHUNK = 64;

__global__ __launch_bounds__(256) void radix_chunk_sums_kernel(const int32_t* __restrict__ hist, int nb, int ntiles, int32_t* __restrict__ chunk_sum)
{
    const int t0 = blockIdx.x * SCAN_CHUNK, t1 = min(t0 + SCAN_CHUNK, ntiles);
    for (int d = threadIdx.x; d < nb; d += 256) {
        int32_t acc = 0;
#pragma unroll 8
        for (int t = t0; t < t1; t++) acc += hist[(int64_t)t * nb + d];
        chunk_sum[(int64_t)blockIdx.x * nb + d] = acc;
    }
}

// one workgroup, one thread per digit: chunk_off[c][d] = sum_{c' < c} chunk_sum[c'][d]; digit_base[d] = sum_{d' < d} total[d']
__global__ __launch_bounds__(1024) void radix_chunk_bases_kernel(const int32_t* __restrict__ chunk_sum, int nb, int nchunks,
                                                                 int32_t* __restrict__ chunk_off, int32_t* __restrict__ digit_base)
{
    __shared__ int32_t s_w[16];
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    int32_t total = 0;
    if (d < nb) {
#pragma unroll 8
        for (int c = 0; c < nchunks; c++) {
            const int32_t v = chunk_sum[(int64_t)c * nb + d];
            chunk_off[(int64_t)c * nb + d] = total;
            total += v;
        }
    }
    int32_t inc = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int32_t base = inc - total;
    for (int w = 0; w < wave; w++) base += s_w[w];
    if (d < nb) digit_base[d] = base;
}

__global__ __launch_bounds__(256) void radix_chunk_apply_kernel(const int32_t* __restrict__ hist, int nb, int ntiles,
                                                                const int32_t* __restrict__ chunk_off, const int32_t* __restrict__ digit_base,
                                                                int32_t* __restrict__ offs)
{
    const int t0 = blockIdx.x * SCAN_CHUNK, t1 = min(t0 + SCAN_CHUNK, ntiles);
    for (int d = threadIdx.x; d < nb; d += 256) {
        int32_t run = chunk_off[(int64_t)blockIdx.x * nb + d] + digit_base[d];
#pragma unroll 8
        for (int t = t0; t < t1; t++) {
            const int32_t v = hist[(int64_t)t * nb + d];
            offs[(int64_t)t * nb + d] = run;
            run += v;
        }
    }
}

// STABLE rank of every element of a wave's segment among the segment's elements of the same digit, one wave instruction (64
// consecutive elements) per call; cnt = the wave's running per-digit counts, two 16-bit counters per LDS word.
//   all valid lanes hold one digit (the usual case for the high digits of coherent input): rank = count so far + lane position.
//   RTN : one LDS atomic-with-return per lane, add 1 to the digit's counter.  Lanes that hit the same counter in one instruction are
//         served in ascending lane order on gfx950 -- a property of the hardware the ISA manual does not promise, so the library checks
//         it on the device before the first sort (rtn_order_probe_kernel) and otherwise takes
//   !RTN: the lane's peers from `bits` ballots; rank = count so far + peers in lower lanes, the lowest peer writes the new count.
template <bool RTN>
__device__ __forceinline__ uint32_t wave_rank(uint32_t* cnt, uint32_t digit, int bits, bool valid, int lane)
{
    const uint64_t vm = __ballot(valid);
    if (vm == 0ull) return 0u;
    const int first = __ffsll((long long)vm) - 1;
    const uint32_t d0 = __builtin_amdgcn_readlane(digit, first);
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t rank = 0u;
    if (__ballot(valid && digit != d0) == 0ull) {
        const uint32_t sh = 16u * (d0 & 1u);
        volatile uint32_t* c = cnt + (d0 >> 1);
        const uint32_t prev = (*c >> sh) & 0xFFFFu;
        __builtin_amdgcn_wave_barrier();
        if (lane == first) *c = *c + ((uint32_t)__popcll(vm) << sh);
        __builtin_amdgcn_wave_barrier();
        rank = prev + (uint32_t)__popcll(vm & lt);
    } else if (RTN) {
        const uint32_t sh = 16u * (digit & 1u);
        if (valid) rank = (atomicAdd(cnt + (digit >> 1), 1u << sh) >> sh) & 0xFFFFu;
    } else {
        const uint64_t peers = digit_peers(digit, bits, valid);
        const uint32_t sh = 16u * (digit & 1u);
        volatile uint32_t* c = cnt + (digit >> 1);
        const uint32_t before = (uint32_t)__popcll(peers & lt);
        const uint32_t prev = valid ? ((*c >> sh) & 0xFFFFu) : 0u;
        __builtin_amdgcn_wave_barrier();
        // the two digits of a word may both be updated in one instruction: atomic add (no order needed among DIFFERENT digits)
        if (valid && before == 0u) atomicAdd(cnt + (digit >> 1), (uint32_t)__popcll(peers) << sh);
        __builtin_amdgcn_wave_barrier();
        rank = prev + before;
    }
    return rank;
}

// one wave, `rounds` instructions of pseudo-random digits of every conflict degree: do the atomic-with-return ranks equal the ballot
// ranks?  out[0] = number of lanes that differ (0 = same-address LDS atomics of one instruction are served in lane order).
__global__ __launch_bounds__(64) void rtn_order_probe_kernel(int rounds, int32_t* __restrict__ out)
{
    __shared__ uint32_t s_a[NBMAX / 2], s_b[NBMAX / 2];
    const int lane = threadIdx.x;
    for (int d = lane; d < NBMAX / 2; d += 64) { s_a[d] = 0u; s_b[d] = 0u; }
    __syncthreads();
    int bad = 0;
    uint32_t state = 0x9E3779B9u * (uint32_t)(lane + 1);
    for (int r = 0; r < rounds; r++) {
        state = state * 1664525u + 1013904223u;
        const int bits = 1 + (r % MAXB);                               // 2 .. 1024 distinct digits: 32-fold conflicts down to none
        const uint32_t digit = (state >> 11) & ((1u << bits) - 1u);
        const bool valid = ((state >> 7) & 31u) != 0u;                 // a few inactive lanes
        const uint32_t ra = wave_rank<true>(s_a, digit, MAXB, valid, lane);
        const uint32_t rb = wave_rank<false>(s_b, digit, MAXB, valid, lane);
        bad += (valid && ra != rb) ? 1 : 0;
        if ((r & 63) == 63) {                                          // keep the 16-bit counters from overflowing
            __syncthreads();
            for (int d = lane; d < NBMAX / 2; d += 64) { s_a[d] = 0u; s_b[d] = 0u; }
            __syncthreads();
        }
    }
    atomicAdd(out, bad);
}

// offs = exclusive scan of hist in digit-major order, stored like hist ([tile][digit]).  Elements travel as (key, index) PAIRS in one
// column: a digit's run of a tile is then one contiguous piece of 8 bytes per element (two separate 4-byte columns made two half
// as long pieces -- with ten-bit digits of incoherent low bits a run averages 8 elements).  FIRST: keys come from the key column of
// pass 0's count kernel and the index of an element is its position; LAST: only the indices are written (the permutation).
template <bool FIRST, bool LAST, bool RTN>
__global__ __launch_bounds__(ST) void radix_scatter_kernel(int64_t n, const uint32_t* __restrict__ keys_in, const uint2* __restrict__ pairs_in,
                                                            uint2* __restrict__ pairs_out, uint32_t* __restrict__ idx_out, int shift, int bits,
                                                            const int32_t* __restrict__ offs, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    uint2* s_pair = reinterpret_cast<uint2*>(s_dyn);      // [TILE]
    // 16-bit counters (a segment has 2048 elements), two per word: 128 KB staging + 22 KB of tables = one workgroup of 512 threads per CU
    __shared__ uint32_t s_cnt[SW][NBMAX / 2];      // per wave: running count of each digit in its segment -> exclusive prefix over the waves
    __shared__ uint16_t s_start[NBMAX];            // first staging slot of each digit
    __shared__ int32_t s_goff[NBMAX];              // global position of the tile's first element of each digit, minus s_start
    __shared__ int32_t s_wsum[SW];
    const int nb = 1 << bits;
    const uint32_t mask = (uint32_t)nb - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < SW * NBMAX / 2; d += ST) (&s_cnt[0][0])[d] = 0u;
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int64_t base = tile0 + (int64_t)wave * SEG + lane;
    uint32_t key[SI], idx[SI];
    uint16_t rank[SI];
#pragma unroll
    for (int r = 0; r < SI; r++) {
        const int64_t i = base + (int64_t)r * 64;
        if (FIRST) {
            key[r] = (i < n) ? keys_in[i] : 0xFFFFFFFFu;
            idx[r] = (uint32_t)i;
        } else {
            const uint2 v = (i < n) ? pairs_in[i] : make_uint2(0xFFFFFFFFu, 0u);
            key[r] = v.x; idx[r] = v.y;
        }
    }
#pragma unroll
    for (int r = 0; r < SI; r++)
        rank[r] = (uint16_t)wave_rank<RTN>(s_cnt[wave], (key[r] >> shift) & mask, bits, base + (int64_t)r * 64 < n, lane);
    __syncthreads();
    // per digit: exclusive prefix of the waves' counts (in place) and the tile total; then the exclusive scan of the totals over the digits
    constexpr int DPT = (NBMAX + ST - 1) / ST;                        // digits per thread: thread t owns [DPT t, DPT t + DPT)
    uint16_t* c16 = reinterpret_cast<uint16_t*>(&s_cnt[0][0]);        // [SW][NBMAX] as 16-bit counters (little endian: digit d at 2 (d >> 1) + (d & 1) = d)
    int32_t tot[DPT];
    int32_t local = 0;
#pragma unroll
    for (int k = 0; k < DPT; k++) {
        const int d = threadIdx.x * DPT + k;
        int32_t run = 0;
        if (d < nb)
#pragma unroll
            for (int w = 0; w < SW; w++) { const int32_t c = c16[w * NBMAX + d]; c16[w * NBMAX + d] = (uint16_t)run; run += c; }
        tot[k] = run;
        local += run;
    }
    int32_t inc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int32_t run = inc - local;
    for (int w = 0; w < wave; w++) run += s_wsum[w];
#pragma unroll
    for (int k = 0; k < DPT; k++) {
        const int d = threadIdx.x * DPT + k;
        if (d < nb) {
            s_start[d] = (uint16_t)run;
            s_goff[d] = offs[(int64_t)blockIdx.x * nb + d] - run;
        }
        run += tot[k];
    }
    __syncthreads();
    // stage the tile in digit order (stable: wave, round, lane)
#pragma unroll
    for (int r = 0; r < SI; r++) {
        if (base + (int64_t)r * 64 < n) {
            const uint32_t digit = (key[r] >> shift) & mask;
            const int pos = (int)s_start[digit] + (int)c16[wave * NBMAX + digit] + (int)rank[r];
            s_pair[pos] = make_uint2(key[r], idx[r]);
        }
    }
    __syncthreads();
    const int count = (int)((n - tile0) < TILE ? (n - tile0) : TILE);
    for (int j = threadIdx.x; j < count; j += ST) {
        const uint2 v = s_pair[j];
        const int64_t dst = (int64_t)s_goff[(v.x >> shift) & mask] + j;
        if (LAST) idx_out[dst] = v.y;
        else pairs_out[dst] = v;
    }
}

struct SortPlan {
    int npass, bits[4], shift[4];
    int ntiles;
    size_t col, hist_bytes, scan_bytes;
};

SortPlan sort_plan(int64_t n, int drop_bits)
{
    SortPlan p;
    const int nbits = 30 - drop_bits;
    p.npass = (nbits + MAXB - 1) / MAXB;
    int left = nbits, sh = drop_bits;
    for (int k = 0; k < p.npass; k++) {
        const int b = (left + (p.npass - k) - 1) / (p.npass - k);      // as even as possible: 10 10 10, 9 9 9, 8 8 8
        p.bits[k] = b; p.shift[k] = sh; sh += b; left -= b;
    }
    p.ntiles = (int)((n + TILE - 1) / TILE);
    p.col = align256((size_t)n * 4);
    p.hist_bytes = align256((size_t)NBMAX * (size_t)p.ntiles * 4);
    p.scan_bytes = align256((size_t)NBMAX * (size_t)(2 * ((p.ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK) + 1) * 4);
    return p;
}

int g_rank_mode = 0;

}  // namespace

// which in-wave ranking ia_morton_order runs: 0 = not decided yet (no sort so far), 1 = LDS atomics with return (the device check passed),
// 2 = ballots (IA_SORT_RANK=ballot, or the check failed)
IA_EXPORT int ia_sort_rank_mode(void) { return g_rank_mode; }

IA_EXPORT size_t ia_morton_order_tmp_bytes(int64_t n)
{
    if (n <= 0) return 256;
    const SortPlan p = sort_plan(n, 0);
    return 4 * p.col + 2 * p.hist_bytes + p.scan_bytes + 256;
}

// order [n] int32: the permutation that lists the points by the Morton code of their cell (cell size 1 / inv_cell, 10 bits per
// axis from `origin`), stable; drop_bits (0, 3 or 6) low key bits are ignored by the sort.
IA_EXPORT int ia_morton_order(int64_t n, const float* pts, const float* origin_host3, float inv_cell, int drop_bits, int32_t* order,
                              void* tmp, size_t tmp_bytes, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n < ((int64_t)1 << 31), "ia_morton_order: n must stay below 2^31");
    IA_REQUIRE(drop_bits >= 0 && drop_bits < 30, "ia_morton_order: drop_bits out of range");
    IA_REQUIRE(tmp != nullptr && tmp_bytes >= ia_morton_order_tmp_bytes(n), "ia_morton_order: tmp too small (ia_morton_order_tmp_bytes)");
    IA_REQUIRE((reinterpret_cast<uintptr_t>(tmp) & 255) == 0, "ia_morton_order: tmp must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const SortPlan p = sort_plan(n, drop_bits);
    char* base = reinterpret_cast<char*>(tmp);
    // two pair columns (8 B per element); the key column of pass 0 lives in the second one, which pass 0 does not write
    uint2* pbuf[2] = {reinterpret_cast<uint2*>(base), reinterpret_cast<uint2*>(base + 2 * p.col)};
    uint32_t* keys0 = reinterpret_cast<uint32_t*>(base + 2 * p.col);
    int32_t* hist = reinterpret_cast<int32_t*>(base + 4 * p.col);
    int32_t* offs = reinterpret_cast<int32_t*>(base + 4 * p.col + p.hist_bytes);
    void* scan_tmp = base + 4 * p.col + 2 * p.hist_bytes;
    static const int raster = getenv("IA_SORT_RASTER_BITS") ? atoi(getenv("IA_SORT_RASTER_BITS")) : 0;       // experiment knob, see morton_key
    const size_t lds = (size_t)2 * TILE * sizeof(uint32_t);
    // once per DEVICE: 2 x TILE x 4 = 128 KB of dynamic LDS per workgroup of the scatter kernels (above the 64 KB a kernel gets without the
    // attribute; + 22 KB static), and the device check of the atomic ranking -- a process that sorts on several devices sets each one up
    // with that device current, and trusts the ranking mode probed on THAT device
    constexpr int MAX_DEVICES = 64;
    static std::once_flag once[MAX_DEVICES];
    static bool rtn_ok_dev[MAX_DEVICES];
    static int setup_rc[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        ia::set_error("ia_morton_order: hipGetDevice failed or device index >= %d", MAX_DEVICES);
        return IA_ERR_LAUNCH;
    }
    std::call_once(once[dev], [lds, dev]() {
        rtn_ok_dev[dev] = false;
        setup_rc[dev] = 0;
#define IA_SET_LDS(F, L, R) if (hipFuncSetAttribute(reinterpret_cast<const void*>(&radix_scatter_kernel<F, L, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) setup_rc[dev] = 1
        IA_SET_LDS(true, false, true); IA_SET_LDS(false, false, true); IA_SET_LDS(false, true, true); IA_SET_LDS(true, true, true);
        IA_SET_LDS(true, false, false); IA_SET_LDS(false, false, false); IA_SET_LDS(false, true, false); IA_SET_LDS(true, true, false);
#undef IA_SET_LDS
        const char* mode = getenv("IA_SORT_RANK");                     // "ballot" forces the fallback (tests run both)
        if (mode && !strcmp(mode, "ballot")) return;
        int32_t* d = nullptr;
        int32_t h = -1;
        if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(int32_t)) != hipSuccess) return;
        if (hipMemset(d, 0, sizeof(int32_t)) == hipSuccess) {
            rtn_order_probe_kernel<<<1, 64, 0, 0>>>(2048, d);
            if (hipMemcpy(&h, d, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) h = -1;
        }
        (void)hipFree(d);
        rtn_ok_dev[dev] = (h == 0);
    });
    if (setup_rc[dev] != 0) {
        (void)hipGetLastError();
        ia::set_error("ia_morton_order: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed on device %d", lds, dev);
        return IA_ERR_LAUNCH;
    }
    const bool rtn_ok = rtn_ok_dev[dev];
    g_rank_mode = rtn_ok ? 1 : 2;                     // diagnostic (ia_sort_rank_mode): the mode of the device that sorted last
    for (int k = 0; k < p.npass; k++) {
        const bool first = k == 0, last = k == p.npass - 1;
        const uint2* pin = first ? nullptr : pbuf[(k - 1) & 1];        // pass k > 0 reads what pass k - 1 wrote
        uint2* pout = pbuf[k & 1];
        const int nb = 1 << p.bits[k];
        if (first)
            radix_count_kernel<true><<<p.ntiles, CT, 0, s>>>(n, pts, origin_host3[0], origin_host3[1], origin_host3[2], inv_cell, keys0, nullptr,
                                                             p.shift[k], p.bits[k], hist, p.ntiles, raster);
        else
            radix_count_kernel<false><<<p.ntiles, CT, 0, s>>>(n, nullptr, 0.f, 0.f, 0.f, 0.f, nullptr, pin, p.shift[k], p.bits[k], hist, p.ntiles, 0);
        const int nchunks = (p.ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
        int32_t* chunk_sum = reinterpret_cast<int32_t*>(scan_tmp);
        int32_t* chunk_off = chunk_sum + (size_t)NBMAX * nchunks;
        int32_t* digit_base = chunk_off + (size_t)NBMAX * nchunks;
        radix_chunk_sums_kernel<<<nchunks, 256, 0, s>>>(hist, nb, p.ntiles, chunk_sum);
        radix_chunk_bases_kernel<<<1, 1024, 0, s>>>(chunk_sum, nb, nchunks, chunk_off, digit_base);
        radix_chunk_apply_kernel<<<nchunks, 256, 0, s>>>(hist, nb, p.ntiles, chunk_off, digit_base, offs);
#define IA_SCATTER(F, L, R) radix_scatter_kernel<F, L, R><<<p.ntiles, ST, lds, s>>>(n, keys0, pin, pout, reinterpret_cast<uint32_t*>(order), p.shift[k], p.bits[k], offs, p.ntiles)
        if (rtn_ok) {
            if (first && last) IA_SCATTER(true, true, true); else if (first) IA_SCATTER(true, false, true);
            else if (last) IA_SCATTER(false, true, true); else IA_SCATTER(false, false, true);
        } else {
            if (first && last) IA_SCATTER(true, true, false); else if (first) IA_SCATTER(true, false, false);
            else if (last) IA_SCATTER(false, true, false); else IA_SCATTER(false, false, false);
        }
#undef IA_SCATTER
    }
    return ia::check_launch("ia_morton_order");
}

IA_EXPORT int ia_gather_rows3_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    gather_rows3_i32_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_gather_rows3_i32");
}

IA_EXPORT int ia_scatter_f32_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    scatter_f32_i32_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_scatter_f32_i32");
}

IA_EXPORT int ia_scatter_rows3_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    scatter_rows3_i32_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_scatter_rows3_i32");
}
