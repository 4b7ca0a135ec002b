// traverse.hip -- occupancy-grid ray marching for gfx950 (replaces nerfacc.traverse_grids;
// reference call sites models/occ_grid/temporal_occ_grid.py:166-175, models/intrinsic_avatar.py:84-93).
//
// Layout / mapping
//   * the 64^3 occupancy grid is bit-packed (32 KiB) and staged ONCE per workgroup into LDS
//     with coalesced 16-byte loads; every cell test of the DDA is then a ds_read_b32 + bit test;
//   * one ray per lane, 256 rays per workgroup; rays of one workgroup are pixel-neighbours, so
//     a wave's lanes walk neighbouring cells (LDS broadcast / few bank conflicts);
//   * two passes with ONE exclusive scan in between (edge and sample counts travel packed in one int64):
//       pass 1 walks the DDA once and records, per ray, its runs of contiguous samples as
//              (t at run start, #samples) descriptors in a scratch buffer (8 runs inline);
//       pass 2 is a pure expansion: it replays t_{k+1} = t_k + dt from each run start (the same float
//              recurrence as the marching loop, so values stay bit-exact) and streams the packed outputs --
//              no second DDA, no LDS grid.  Rays with more than 8 runs (rare) re-walk the DDA in pass 2.
//     Output order is ray order then marching order, exactly as upstream.
// Arithmetic is kept operation-for-operation identical to oracle/ia_oracle.c (this TU is
// built with -ffp-contract=off) so edge/sample counts and t values are bit-exact.
#include <stdlib.h>

#include "ia_common.h"
#include "ia_zero.h"
#include "t_advance.h"

namespace {

constexpr int TR_THREADS = 256;
constexpr int TR_RUNS = 8;          // run descriptors kept inline per ray

struct RayScratch {                 // 80 bytes per ray
    float t_first[TR_RUNS];
    int32_t n_samples[TR_RUNS];
    int32_t n_runs;                 // may exceed TR_RUNS (=> pass 2 re-walks this ray)
    float t_term;                   // termination plane
    int32_t pad[2];
};

__device__ __forceinline__ float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    float v = t * cone_angle;
    return v < dt_min ? dt_min : (v > dt_max ? dt_max : v);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ bool ray_aabb(const float o[3], const float d[3], const float* aabb, float& tmin,
                                         float& tmax)
{
    float tmin_t, tmax_t;
    if (d[0] >= 0) { tmin = (aabb[0] - o[0]) / d[0]; tmax = (aabb[3] - o[0]) / d[0]; }
    else           { tmin = (aabb[3] - o[0]) / d[0]; tmax = (aabb[0] - o[0]) / d[0]; }
    if (d[1] >= 0) { tmin_t = (aabb[1] - o[1]) / d[1]; tmax_t = (aabb[4] - o[1]) / d[1]; }
    else           { tmin_t = (aabb[4] - o[1]) / d[1]; tmax_t = (aabb[1] - o[1]) / d[1]; }
    if (tmin > tmax_t || tmin_t > tmax) return false;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (d[2] >= 0) { tmin_t = (aabb[2] - o[2]) / d[2]; tmax_t = (aabb[5] - o[2]) / d[2]; }
    else           { tmin_t = (aabb[5] - o[2]) / d[2]; tmax_t = (aabb[2] - o[2]) / d[2]; }
    if (tmin > tmax_t || tmin_t > tmax) return false;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (tmax <= 0) return false;
    return true;
}

__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t* __restrict__ binaries, int64_t n_cells,
                                                        uint32_t* __restrict__ bits)
{
    // one lane per cell; a wave's 64-bit ballot is two output words
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool occ = (c < n_cells) && binaries[c] != 0;
    const unsigned long long m = __ballot(occ);
    const int lane = threadIdx.x & 63;
    const int64_t w = c >> 5;
    const int64_t n_words = (n_cells + 31) >> 5;
    if (lane == 0 && w < n_words) bits[w] = (uint32_t)m;
    if (lane == 32 && w < n_words) bits[w] = (uint32_t)(m >> 32);
}

template <bool FIRST_PASS>
__global__ __launch_bounds__(TR_THREADS) void traverse_kernel(
    int64_t n_rays, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const uint32_t* __restrict__ grid_bits, int rx, int ry, int rz, const float* __restrict__ aabb_g,
    const float* __restrict__ near_planes, const float* __restrict__ far_planes, float step_size, float cone_angle,
    RayScratch* __restrict__ scratch, int64_t* __restrict__ packed_cnt, bool only_overflow,
    const int64_t* __restrict__ packed_start,
    float* __restrict__ iv_vals, uint8_t* __restrict__ iv_is_left, uint8_t* __restrict__ iv_is_right,
    int64_t* __restrict__ iv_ray, float* __restrict__ sm_vals, int64_t* __restrict__ sm_ray,
    float* __restrict__ term_planes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
    const int n_words = (rx * ry * rz + 31) >> 5;
    // pass 2 only needs the grid for rays whose runs overflowed the inline descriptors
    if (!FIRST_PASS) {
        const int64_t t0 = (int64_t)blockIdx.x * TR_THREADS + threadIdx.x;
        const bool ovf = t0 < n_rays && scratch[t0].n_runs > TR_RUNS;
        if (!__syncthreads_or(ovf)) return;
    }
    // stage the bit grid: 16 B per lane per step when the word count allows
    {
        const int n_vec = n_words >> 2;
        const uint4* src = reinterpret_cast<const uint4*>(grid_bits);
        uint4* dst = reinterpret_cast<uint4*>(s_bits);
        for (int i = threadIdx.x; i < n_vec; i += TR_THREADS) dst[i] = src[i];
        for (int i = (n_vec << 2) + threadIdx.x; i < n_words; i += TR_THREADS) s_bits[i] = grid_bits[i];
    }
    __syncthreads();

    const int64_t tid = (int64_t)blockIdx.x * TR_THREADS + threadIdx.x;
    if (tid >= n_rays) return;
    if (!FIRST_PASS && scratch[tid].n_runs <= TR_RUNS) return;       // expanded by expand_kernel

    float aabb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) aabb[k] = aabb_g[k];
    const int res[3] = {rx, ry, rz};
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { o[k] = rays_o[tid * 3 + k]; d[k] = rays_d[tid * 3 + k]; }
    const float near_plane = near_planes[tid], far_plane = far_planes[tid];

    int64_t n_samples = 0, n_intervals = 0;
    int64_t iv_base = 0, sm_base = 0;
    if (!FIRST_PASS) { const int64_t ps = packed_start[tid]; iv_base = ps & 0xFFFFFFFFll; sm_base = ps >> 32; }
    int n_runs = 0, run_len = 0;
    RayScratch sc;
    bool continuous = false;
    float t_last = near_plane;
    float tmin, tmax;
    const float eps = 1e-6f;

    if (ray_aabb(o, d, aabb, tmin, tmax)) {
        const float this_tmin = fmaxf(tmin, near_plane);
        const float this_tmax = fminf(tmax, far_plane);
        if (this_tmin < this_tmax) {
            if (step_size <= 0.0f) t_last = this_tmin;
            else for (;;) {
                float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                if (t_last + dt * 0.5f >= this_tmin) break;
                t_last += dt;
            }
            float tdist[3], delta[3];
            int cur[3], stp[3], ovf[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float vs = (aabb[3 + a] - aabb[a]) / (float)res[a];
                const float rs = o[a] + d[a] * (this_tmin + eps);
                const float re = o[a] + d[a] * (this_tmax - eps);
                cur[a] = clampi((int)((rs - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
                const int fin = clampi((int)((re - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
                const int start_index = cur[a] + (d[a] > 0 ? 1 : 0);
                const float tmax_a = ((aabb[a] + ((float)start_index * vs - rs)) / d[a]) + this_tmin;
                const float sf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
                tdist[a] = (d[a] == 0.0f) ? this_tmax : tmax_a;
                stp[a] = (int)sf;
                delta[a] = (d[a] == 0.0f) ? this_tmax : vs / d[a] * sf;
                ovf[a] = fin + stp[a];
            }
            for (;;) {
                float t_traverse = fminf(tdist[0], fminf(tdist[1], tdist[2]));
                t_traverse = fminf(t_traverse, this_tmax);
                const int cell = (cur[0] * ry + cur[1]) * rz + cur[2];
                const bool occ = (s_bits[cell >> 5] >> (cell & 31)) & 1u;
                if (!occ) {
                    if (step_size <= 0.0f) t_last = t_traverse;
                    else for (;;) {
                        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_traverse) break;
                        t_last += dt;
                    }
                    continuous = false;
                } else {
                    for (;;) {
                        float t_next;
                        if (step_size <= 0.0f) t_next = t_traverse;
                        else {
                            float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                            if (t_last + dt * 0.5f >= t_traverse) break;
                            t_next = t_last + dt;
                        }
                        if (!continuous) {
                            if (!FIRST_PASS) {
                                const int64_t idx = iv_base + n_intervals;
                                iv_vals[idx] = t_last; iv_ray[idx] = tid; iv_is_left[idx] = 1;
                                iv_vals[idx + 1] = t_next; iv_ray[idx + 1] = tid; iv_is_right[idx + 1] = 1;
                            } else {
                                if (n_runs > 0 && n_runs <= TR_RUNS) sc.n_samples[n_runs - 1] = run_len;
                                if (n_runs < TR_RUNS) sc.t_first[n_runs] = t_last;
                                n_runs++;
                                run_len = 0;
                            }
                            n_intervals += 2;
                        } else {
                            if (!FIRST_PASS) {
                                const int64_t idx = iv_base + n_intervals;
                                iv_vals[idx] = t_next; iv_ray[idx] = tid;
                                iv_is_left[idx - 1] = 1; iv_is_right[idx] = 1;
                            }
                            n_intervals++;
                        }
                        if (!FIRST_PASS) {
                            const int64_t idx = sm_base + n_samples;
                            sm_vals[idx] = (t_next + t_last) * 0.5f; sm_ray[idx] = tid;
                        }
                        n_samples++;
                        run_len++;
                        continuous = true;
                        t_last = t_next;
                        if (t_next >= t_traverse) break;
                    }
                }
                int a;
                if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) a = 0;
                else if (tdist[1] < tdist[2]) a = 1;
                else a = 2;
                bool done;
                // keep cur/tdist in registers: static indexing only
                if (a == 0) { cur[0] += stp[0]; tdist[0] += delta[0]; done = cur[0] == ovf[0]; }
                else if (a == 1) { cur[1] += stp[1]; tdist[1] += delta[1]; done = cur[1] == ovf[1]; }
                else { cur[2] += stp[2]; tdist[2] += delta[2]; done = cur[2] == ovf[2]; }
                if (done) break;
            }
        }
    }
    if (FIRST_PASS) {
        if (n_runs > 0 && n_runs <= TR_RUNS) sc.n_samples[n_runs - 1] = run_len;
        sc.n_runs = n_runs;
        sc.t_term = t_last;
        RayScratch* dst = scratch + tid;
#pragma unroll
        for (int k = 0; k < TR_RUNS; k++)
            if (k < n_runs) { dst->t_first[k] = sc.t_first[k]; dst->n_samples[k] = sc.n_samples[k]; }
        dst->n_runs = n_runs;
        dst->t_term = t_last;
        packed_cnt[tid] = (int64_t)n_intervals | ((int64_t)n_samples << 32);
    }
}

// pass 2: expansion of the run descriptors (no DDA).  One ray per lane.
__global__ __launch_bounds__(TR_THREADS) void expand_kernel(
    int64_t n_rays, const RayScratch* __restrict__ scratch, const int64_t* __restrict__ packed_cnt,
    const int64_t* __restrict__ packed_start, float step_size, float cone_angle,
    int64_t* __restrict__ iv_pinfo, int64_t* __restrict__ sm_pinfo, float* __restrict__ iv_vals,
    uint8_t* __restrict__ iv_is_left, uint8_t* __restrict__ iv_is_right, int64_t* __restrict__ iv_ray,
    float* __restrict__ sm_vals, int64_t* __restrict__ sm_ray, float* __restrict__ term_planes)
{
    const int64_t tid = (int64_t)blockIdx.x * TR_THREADS + threadIdx.x;
    if (tid >= n_rays) return;
    const int64_t pc = packed_cnt[tid], ps = packed_start[tid];
    int64_t iv = ps & 0xFFFFFFFFll, sm = ps >> 32;
    if (iv_pinfo) { iv_pinfo[2 * tid] = iv; iv_pinfo[2 * tid + 1] = pc & 0xFFFFFFFFll; }
    if (sm_pinfo) { sm_pinfo[2 * tid] = sm; sm_pinfo[2 * tid + 1] = pc >> 32; }
    const RayScratch* sc = scratch + tid;
    if (term_planes) term_planes[tid] = sc->t_term;
    const int n_runs = sc->n_runs;
    if (n_runs > TR_RUNS) return;                                   // re-walked by traverse_kernel<false>
    for (int r = 0; r < n_runs; r++) {
        float t = sc->t_first[r];
        const int n = sc->n_samples[r];
        iv_vals[iv] = t; iv_ray[iv] = tid; iv_is_left[iv] = 1;      // flags are caller-zeroed
        iv++;
        for (int k = 0; k < n; k++) {
            const float t_next = step_size <= 0.0f ? t : t + calc_dt(t, cone_angle, step_size, 1e10f);
            iv_vals[iv] = t_next; iv_ray[iv] = tid; iv_is_right[iv] = 1;
            if (k + 1 < n) iv_is_left[iv] = 1;
            iv++;
            sm_vals[sm] = (t_next + t) * 0.5f; sm_ray[sm] = tid;
            sm++;
            t = t_next;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Single-pass traversal (count + scan + fill in ONE launch) for callers that can bound the output size.
//
//   * tiles of 256 consecutive rays; the tile id is a ticket taken at workgroup start, so a tile's predecessors are
//     always resident or finished (forward progress for the look-back below);
//   * phase 1: the DDA walk of pass 1, run descriptors go to LDS (SoA, conflict-free) instead of HBM;
//   * phase 2: workgroup scan of the packed (edges | samples << 32) counts -> local offsets;
//   * phase 3: decoupled look-back over one 64-bit word per tile {flag:2, samples:31, edges:31}: wave 0 inspects 64
//     predecessors per step and publishes the tile's inclusive prefix -- the global offsets without a second kernel;
//   * phase 4: element-parallel expansion: one lane per output EDGE (binary search of the owning ray in the LDS
//     offsets, replay of t_{k+1} = t_k + dt from the run start -- the marching recurrence, bit-exact), so every
//     store instruction of a wave covers 64 consecutive elements: coalesced 256/512-byte writes.
// HBM traffic = 32 B/ray in + the outputs; no scratch descriptors, no zero-fill of the flag arrays.
constexpr uint64_t TS_AGG = 1ull << 62, TS_PREFIX = 2ull << 62, TS_MASK = (1ull << 62) - 1;
constexpr int FT_OFFS = TR_THREADS + 1;

struct LdsRunSink {
    float* tfirst;      // [TR_RUNS][TR_THREADS]
    int* nsamp;         // [TR_RUNS][TR_THREADS]
    int lane;
    int n_runs = 0, run_len = 0, n_samples = 0, n_intervals = 0;
    __device__ __forceinline__ void emit(float t_last, float, bool continuous)
    {
        if (!continuous) {
            if (n_runs > 0 && n_runs <= TR_RUNS) nsamp[(n_runs - 1) * TR_THREADS + lane] = run_len;
            if (n_runs < TR_RUNS) tfirst[n_runs * TR_THREADS + lane] = t_last;
            n_runs++;
            run_len = 0;
            n_intervals += 2;
        } else {
            n_intervals++;
        }
        n_samples++;
        run_len++;
    }
    __device__ __forceinline__ void finish()
    {
        if (n_runs > 0 && n_runs <= TR_RUNS) nsamp[(n_runs - 1) * TR_THREADS + lane] = run_len;
    }
};

struct DirectWriteSink {       // in-order writes of one ray (rays with more than TR_RUNS runs); writes every flag
    float* iv_vals; uint8_t* iv_is_left; uint8_t* iv_is_right; int64_t* iv_ray; float* sm_vals; int64_t* sm_ray;
    int64_t iv_base, sm_base, tid;
    float* sm_ts = nullptr; float* sm_te = nullptr;        // optional: interval ends per sample (= vals[is_left] / vals[is_right])
    int64_t n_samples = 0, n_intervals = 0;
    __device__ __forceinline__ void emit(float t_last, float t_next, bool continuous)
    {
        const int64_t idx = iv_base + n_intervals;
        if (!continuous) {
            iv_vals[idx] = t_last; iv_ray[idx] = tid; iv_is_left[idx] = 1; iv_is_right[idx] = 0;
            iv_vals[idx + 1] = t_next; iv_ray[idx + 1] = tid; iv_is_left[idx + 1] = 0; iv_is_right[idx + 1] = 1;
            n_intervals += 2;
        } else {
            iv_vals[idx] = t_next; iv_ray[idx] = tid; iv_is_left[idx] = 0; iv_is_right[idx] = 1;
            iv_is_left[idx - 1] = 1;
            n_intervals++;
        }
        const int64_t si = sm_base + n_samples;
        sm_vals[si] = (t_next + t_last) * 0.5f; sm_ray[si] = tid;
        if (sm_ts) { sm_ts[si] = t_last; sm_te[si] = t_next; }
        n_samples++;
    }
};

// the marching loop of traverse_kernel, operation for operation, with the per-sample action factored out
template <class Sink>
__device__ __forceinline__ float dda_walk(const float o[3], const float d[3], const float aabb[6], int rx, int ry, int rz,
                                          float near_plane, float far_plane, float step_size, float cone_angle,
                                          const uint32_t* s_bits, Sink& sink)
{
    const int res[3] = {rx, ry, rz};
    bool continuous = false;
    float t_last = near_plane;
    float tmin, tmax;
    const float eps = 1e-6f;
    if (!ray_aabb(o, d, aabb, tmin, tmax)) return t_last;
    const float this_tmin = fmaxf(tmin, near_plane);
    const float this_tmax = fminf(tmax, far_plane);
    if (!(this_tmin < this_tmax)) return t_last;
    for (;;) {
        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
        if (t_last + dt * 0.5f >= this_tmin) break;
        t_last += dt;
    }
    float tdist[3], delta[3];
    int cur[3], stp[3], ovf[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float vs = (aabb[3 + a] - aabb[a]) / (float)res[a];
        const float rs = o[a] + d[a] * (this_tmin + eps);
        const float re = o[a] + d[a] * (this_tmax - eps);
        cur[a] = clampi((int)((rs - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
        const int fin = clampi((int)((re - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
        const int start_index = cur[a] + (d[a] > 0 ? 1 : 0);
        const float tmax_a = ((aabb[a] + ((float)start_index * vs - rs)) / d[a]) + this_tmin;
        const float sf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
        tdist[a] = (d[a] == 0.0f) ? this_tmax : tmax_a;
        stp[a] = (int)sf;
        delta[a] = (d[a] == 0.0f) ? this_tmax : vs / d[a] * sf;
        ovf[a] = fin + stp[a];
    }
    for (;;) {
        float t_traverse = fminf(tdist[0], fminf(tdist[1], tdist[2]));
        t_traverse = fminf(t_traverse, this_tmax);
        const int cell = (cur[0] * ry + cur[1]) * rz + cur[2];
        const bool occ = (s_bits[cell >> 5] >> (cell & 31)) & 1u;
        if (!occ) {
            for (;;) {
                float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                if (t_last + dt * 0.5f >= t_traverse) break;
                t_last += dt;
            }
            continuous = false;
        } else {
            for (;;) {
                float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                if (t_last + dt * 0.5f >= t_traverse) break;
                const float t_next = t_last + dt;
                sink.emit(t_last, t_next, continuous);
                continuous = true;
                t_last = t_next;
                if (t_next >= t_traverse) break;
            }
        }
        int a;
        if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) a = 0;
        else if (tdist[1] < tdist[2]) a = 1;
        else a = 2;
        bool done;
        if (a == 0) { cur[0] += stp[0]; tdist[0] += delta[0]; done = cur[0] == ovf[0]; }
        else if (a == 1) { cur[1] += stp[1]; tdist[1] += delta[1]; done = cur[1] == ovf[1]; }
        else { cur[2] += stp[2]; tdist[2] += delta[2]; done = cur[2] == ovf[2]; }
        if (done) break;
    }
    return t_last;
}


// cone_angle == 0 specialisation of dda_walk (dt == step_size for every sample), restructured to cut the instruction
// count of the EMPTY part of the grid, where rays spend most of their cells:
//   * marching through an empty cell only moves t_last up to that cell's exit time, and "march until mid >= A, then
//     until mid >= B" equals "march until mid >= max(A, B)": the catch-up is deferred to the next occupied cell (or the
//     end of the ray), so an empty cell costs one DDA step + one LDS bit test and nothing else;
//   * long catch-ups (the stretch in front of the box: ~120 steps for a primary ray) jump with ia_advance() -- the
//     exact closed form of the recurrence -- to a conservative k0 <= k*, verified, then finish with the plain loop;
//   * the DDA step is predicated (selects, no divergent three-way branch);
//   * `if (t_next >= t_traverse) break` of the reference loop is implied by its loop condition (t_next + dt/2 >= t_next).
// Same float operations on the same operands in the same order for everything that reaches an output.
__device__ __forceinline__ float catch_up(float t, float step, float half, float target)
{
    if (!(t + half >= target)) {
        const float est = (target - half - t) / step;
        if (est > 24.0f && est < 1.0e9f) {
            const int e = (int)est;
            const int k0 = e - 2 - (e >> 14);
            const float tj = ia_advance(t, step, k0);
            if (!(tj + half >= target)) t = tj;                     // k0 <= k*: safe to continue from there
        }
        while (!(t + half >= target)) t += step;
    }
    return t;
}

template <class Sink>
__device__ __forceinline__ float dda_walk_fast(const float o[3], const float d[3], const float aabb[6], int rx, int ry,
                                               int rz, float near_plane, float far_plane, float step, const uint32_t* s_bits,
                                               Sink& sink, const int* occ_box = nullptr /* LDS: cell box of the occupied cells, lo[3] hi[3] */)
{
    const int res[3] = {rx, ry, rz};
    float t_last = near_plane;
    float tmin, tmax;
    const float eps = 1e-6f;
    if (!ray_aabb(o, d, aabb, tmin, tmax)) return t_last;
    const float this_tmin = fmaxf(tmin, near_plane);
    const float this_tmax = fminf(tmax, far_plane);
    if (!(this_tmin < this_tmax)) return t_last;
    const float half = step * 0.5f;
    float tdist[3], delta[3];
    int cur[3], stp[3], ovf[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float vs = (aabb[3 + a] - aabb[a]) / (float)res[a];
        const float rs = o[a] + d[a] * (this_tmin + eps);
        const float re = o[a] + d[a] * (this_tmax - eps);
        cur[a] = clampi((int)((rs - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
        const int fin = clampi((int)((re - aabb[a]) / (aabb[3 + a] - aabb[a]) * (float)res[a]), 0, res[a] - 1);
        const int start_index = cur[a] + (d[a] > 0 ? 1 : 0);
        const float tmax_a = ((aabb[a] + ((float)start_index * vs - rs)) / d[a]) + this_tmin;
        const float sf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
        tdist[a] = (d[a] == 0.0f) ? this_tmax : tmax_a;
        stp[a] = (int)sf;
        delta[a] = (d[a] == 0.0f) ? this_tmax : vs / d[a] * sf;
        ovf[a] = fin + stp[a];
    }
    if (occ_box) {
        // Nothing is emitted outside the box of the OCCUPIED cells: a ray that is beyond it on some axis and does not move back can stop,
        // and the walk of every other ray ends when it steps out of it (the per-axis end index is tightened: no extra test in the cell loop).
        // Only the termination plane -- t_last after the last, empty cells -- needs the rest of the walk: callers that want it pass no box.
        bool dead = false;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int lo = occ_box[a], hi = occ_box[3 + a];
            const int last = ovf[a] - stp[a];                 // the walk visits cur[a] .. last on this axis, monotonically
            dead = dead || (stp[a] >= 0 ? (cur[a] > hi || last < lo) : (cur[a] < lo || last > hi));
            if (stp[a] > 0) ovf[a] = min(ovf[a], hi + 1);
            else if (stp[a] < 0) ovf[a] = max(ovf[a], lo - 1);
        }
        if (dead) return t_last;
    }
    // the stretch in front of the box: every lane does it here, together (closed-form jump), so the divergent code in
    // the cell loop below stays tiny; `pending` = max of the deferred empty-cell thresholds inside the box
    t_last = catch_up(t_last, step, half, this_tmin);
    float pending = this_tmin;
    bool continuous = false;
    // linear cell index kept incrementally (integer multiplies are quarter rate)
    int cell = (cur[0] * ry + cur[1]) * rz + cur[2];
    const int cs0 = stp[0] * ry * rz, cs1 = stp[1] * rz, cs2 = stp[2];
    for (;;) {
        float t_traverse = fminf(tdist[0], fminf(tdist[1], tdist[2]));
        t_traverse = fminf(t_traverse, this_tmax);
        const bool occ = (s_bits[cell >> 5] >> (cell & 31)) & 1u;
        if (occ) {
            while (!(t_last + half >= pending)) t_last += step;
            while (!(t_last + half >= t_traverse)) {
                const float t_next = t_last + step;
                sink.emit(t_last, t_next, continuous);
                continuous = true;
                t_last = t_next;
            }
        } else {
            pending = fmaxf(pending, t_traverse);
            continuous = false;
        }
        const bool a0 = (tdist[0] < tdist[1]) && (tdist[0] < tdist[2]);
        const bool a1 = !a0 && (tdist[1] < tdist[2]);
        const bool a2 = !a0 && !a1;
        const float n0 = tdist[0] + delta[0], n1 = tdist[1] + delta[1], n2 = tdist[2] + delta[2];
        tdist[0] = a0 ? n0 : tdist[0]; tdist[1] = a1 ? n1 : tdist[1]; tdist[2] = a2 ? n2 : tdist[2];
        cur[0] += a0 ? stp[0] : 0; cur[1] += a1 ? stp[1] : 0; cur[2] += a2 ? stp[2] : 0;
        cell += a0 ? cs0 : (a1 ? cs1 : cs2);
        const bool done = a0 ? (cur[0] == ovf[0]) : (a1 ? (cur[1] == ovf[1]) : (cur[2] == ovf[2]));
        if (done) break;
    }
    while (!(t_last + half >= pending)) t_last += step;
    return t_last;
}

// cell box of the occupied cells of the staged bit grid -> s_box[6] = lo x, y, z, hi x, y, z (lo > hi: nothing is occupied).  A word holds 32
// consecutive z of one (x, y) when rz is a multiple of 32 (the reference's 64^3 grids); other shapes keep the whole grid.  Call between
// two barriers: after the grid is staged, before the first walk.
__device__ __forceinline__ void occupied_cell_box(const uint32_t* s_bits, int rx, int ry, int rz, int* s_box, int tid, int nthreads)
{
    if (tid < 3) s_box[tid] = 0x7fffffff;
    else if (tid < 6) s_box[tid] = -1;
    __syncthreads();
    if ((rz & 31) != 0) {
        if (tid == 0) { s_box[0] = 0; s_box[1] = 0; s_box[2] = 0; s_box[3] = rx - 1; s_box[4] = ry - 1; s_box[5] = rz - 1; }
        return;
    }
    const int n_words = (rx * ry * rz) >> 5, wz = rz >> 5;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    for (int w = tid; w < n_words; w += nthreads) {
        const uint32_t v = s_bits[w];
        if (v == 0u) continue;
        const int col = w / wz, zb = (w - col * wz) << 5;
        const int x = col / ry, y = col - x * ry;
        lo[0] = min(lo[0], x); hi[0] = max(hi[0], x);
        lo[1] = min(lo[1], y); hi[1] = max(hi[1], y);
        lo[2] = min(lo[2], zb + __builtin_ctz(v)); hi[2] = max(hi[2], zb + 31 - __builtin_clz(v));
    }
    if (hi[0] >= 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { atomicMin(&s_box[a], lo[a]); atomicMax(&s_box[3 + a], hi[a]); }
    }
}

__device__ __forceinline__ uint64_t ts_load(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ts_store(uint64_t* p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(TR_THREADS) void traverse_fused_kernel(
    int64_t n_rays, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const uint32_t* __restrict__ grid_bits, int rx, int ry, int rz, const float* __restrict__ aabb_g,
    const float* __restrict__ near_planes, const float* __restrict__ far_planes, float step_size, float cone_angle,
    uint64_t* __restrict__ tile_state /*[n_tiles] zeroed*/, uint32_t* __restrict__ ticket /*zeroed*/, int n_tiles,
    int64_t cap_edges, int64_t cap_samples, int64_t* __restrict__ totals /*[3]: edges, samples, overflow*/,
    int64_t* __restrict__ iv_pinfo, int64_t* __restrict__ sm_pinfo, float* __restrict__ iv_vals,
    uint8_t* __restrict__ iv_is_left, uint8_t* __restrict__ iv_is_right, int64_t* __restrict__ iv_ray,
    float* __restrict__ sm_vals, int64_t* __restrict__ sm_ray, float* __restrict__ term_planes,
    float* __restrict__ sm_ts /*or NULL*/, float* __restrict__ sm_te /*or NULL*/)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
    __shared__ float s_tfirst[TR_RUNS * TR_THREADS];
    __shared__ int s_nsamp[TR_RUNS * TR_THREADS];
    __shared__ int s_nruns[TR_THREADS];
    __shared__ int s_offE[FT_OFFS], s_offS[FT_OFFS];
    __shared__ unsigned long long s_wave_tot[TR_THREADS / 64];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_tile;
    const int lane_wg = threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    {
        const int n_words = (rx * ry * rz + 31) >> 5;
        const int n_vec = n_words >> 2;
        const uint4* src = reinterpret_cast<const uint4*>(grid_bits);
        uint4* dst = reinterpret_cast<uint4*>(s_bits);
        for (int i = threadIdx.x; i < n_vec; i += TR_THREADS) dst[i] = src[i];
        for (int i = (n_vec << 2) + threadIdx.x; i < n_words; i += TR_THREADS) s_bits[i] = grid_bits[i];
    }
    float aabb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) aabb[k] = aabb_g[k];
    __shared__ int s_box[6];
    __syncthreads();
    occupied_cell_box(s_bits, rx, ry, rz, s_box, threadIdx.x, TR_THREADS);
    const int* const occ_box = term_planes ? nullptr : s_box;
    // persistent workgroups: the bit grid is staged once, tiles are pulled off the ticket counter until none is left
    // (per-tile staging + workgroup launch were 141 us of the 674 us of a 2 M-ray batch)
  for (;;) {
    __syncthreads();                       // previous tile fully written; LDS descriptors free
    if (lane_wg == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = (int)s_tile;
    if (tile >= n_tiles) break;
    const int64_t tid = (int64_t)tile * TR_THREADS + lane_wg;
    const bool active = tid < n_rays;

    // ---- phase 1: walk
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f};
    float near_plane = 0.f, far_plane = 0.f, t_term = 0.f;
    LdsRunSink sink{s_tfirst, s_nsamp, lane_wg};
    if (active) {
#pragma unroll
        for (int k = 0; k < 3; k++) { o[k] = rays_o[tid * 3 + k]; d[k] = rays_d[tid * 3 + k]; }
        near_plane = near_planes[tid]; far_plane = far_planes[tid];
        t_term = (cone_angle == 0.0f)
                     ? dda_walk_fast(o, d, aabb, rx, ry, rz, near_plane, far_plane, step_size, s_bits, sink, occ_box)
                     : dda_walk(o, d, aabb, rx, ry, rz, near_plane, far_plane, step_size, cone_angle, s_bits, sink);
        sink.finish();
    }
    s_nruns[lane_wg] = sink.n_runs;

    // ---- phase 2: workgroup exclusive scan of the packed counts
    const unsigned long long mine = (unsigned long long)sink.n_intervals | ((unsigned long long)sink.n_samples << 32);
    unsigned long long inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long v = __shfl_up(inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave_tot[wid] = inc;
    __syncthreads();
    unsigned long long wave_off = 0, wg_total = 0;
#pragma unroll
    for (int w = 0; w < TR_THREADS / 64; w++) {
        const unsigned long long t = s_wave_tot[w];
        if (w < wid) wave_off += t;
        wg_total += t;
    }
    const unsigned long long excl = wave_off + inc - mine;
    s_offE[lane_wg] = (int)(excl & 0xFFFFFFFFull);
    s_offS[lane_wg] = (int)(excl >> 32);
    if (lane_wg == 0) { s_offE[TR_THREADS] = (int)(wg_total & 0xFFFFFFFFull); s_offS[TR_THREADS] = (int)(wg_total >> 32); }
    const int E_wg = (int)(wg_total & 0xFFFFFFFFull), S_wg = (int)(wg_total >> 32);

    // ---- phase 3: decoupled look-back (wave 0); word = flag | samples << 31 | edges
    if (wid == 0) {
        const uint64_t agg = (uint64_t)E_wg | ((uint64_t)S_wg << 31);
        uint64_t exclusive = 0;
        if (tile > 0) {
            if (lane == 0) ts_store(tile_state + tile, TS_AGG | agg);
            int pos = tile - 1;
            for (;;) {
                const int idx = pos - lane;
                uint64_t v = TS_PREFIX;                              // virtual tiles before tile 0: prefix 0
                if (idx >= 0) {
                    v = ts_load(tile_state + idx);
                    while ((v >> 62) == 0) { __builtin_amdgcn_s_sleep(1); v = ts_load(tile_state + idx); }
                }
                const unsigned long long is_p = __ballot((v >> 62) == 2);
                const int first_p = is_p ? __builtin_ctzll(is_p) : 64;
                uint64_t val = (lane <= first_p) ? (v & TS_MASK) : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
                exclusive += val;
                if (is_p) break;
                pos -= 64;
            }
        }
        if (lane == 0) {
            ts_store(tile_state + tile, TS_PREFIX | (exclusive + agg));
            s_base = exclusive;
            if (tile == n_tiles - 1) {
                const uint64_t tot = exclusive + agg;
                totals[0] = (int64_t)(tot & 0x7FFFFFFFull);
                totals[1] = (int64_t)(tot >> 31);
            }
        }
    }
    __syncthreads();
    const int64_t baseE = (int64_t)(s_base & 0x7FFFFFFFull), baseS = (int64_t)(s_base >> 31);
    if (baseE + E_wg > cap_edges || baseS + S_wg > cap_samples) {
        if (lane_wg == 0) totals[2] = 1;
        continue;
    }

    // ---- phase 4: per-ray records, then element-parallel expansion
    if (active) {
        if (iv_pinfo) { iv_pinfo[2 * tid] = baseE + s_offE[lane_wg]; iv_pinfo[2 * tid + 1] = sink.n_intervals; }
        if (sm_pinfo) { sm_pinfo[2 * tid] = baseS + s_offS[lane_wg]; sm_pinfo[2 * tid + 1] = sink.n_samples; }
        if (term_planes) term_planes[tid] = t_term;
    }
    for (int j = lane_wg; j < E_wg; j += TR_THREADS) {
        int lo = 0, hi = TR_THREADS;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int mid = (lo + hi) >> 1;
            if (s_offE[mid] <= j) lo = mid; else hi = mid;
        }
        const int r = lo;
        if (s_nruns[r] > TR_RUNS) continue;                         // written in order by its own lane below
        int k = j - s_offE[r];
        int sidx = s_offS[r];
        int qn = 0;
        int n = s_nsamp[r];
        while (k > n) { k -= n + 1; sidx += n; qn++; n = s_nsamp[qn * TR_THREADS + r]; }
        float t = s_tfirst[qn * TR_THREADS + r];
        if (cone_angle == 0.0f) t = ia_advance(t, step_size, k);     // exact closed form of the k-fold recurrence (dt == step)
        else for (int i = 0; i < k; i++) t = t + calc_dt(t, cone_angle, step_size, 1e10f);
        const int64_t ray = (int64_t)tile * TR_THREADS + r;
        const int64_t gi = baseE + j;
        iv_vals[gi] = t; iv_ray[gi] = ray; iv_is_left[gi] = k < n; iv_is_right[gi] = k > 0;
        if (k < n) {
            const float t_next = t + calc_dt(t, cone_angle, step_size, 1e10f);
            const int64_t gs = baseS + sidx + k;
            sm_vals[gs] = (t_next + t) * 0.5f; sm_ray[gs] = ray;
            if (sm_ts) { sm_ts[gs] = t; sm_te[gs] = t_next; }      // t_next IS the next edge's value (ia_advance is the exact recurrence)
        }
    }
    if (active && sink.n_runs > TR_RUNS) {
        DirectWriteSink w{iv_vals, iv_is_left, iv_is_right, iv_ray, sm_vals, sm_ray, baseE + s_offE[lane_wg],
                          baseS + s_offS[lane_wg], tid, sm_ts, sm_te};
        (void)dda_walk(o, d, aabb, rx, ry, rz, near_plane, far_plane, step_size, cone_angle, s_bits, w);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The fused traversal with the rays of a tile walked in order of their SPAN inside the grid's box.
// traverse_fused_kernel gives lane l of a 256-ray tile ray l: counters (profiles/r02_traverse_pmc.txt) show 34.7 of 64 lanes
// active per VALU instruction -- a wave runs until its longest ray is through, and neighbouring secondary rays (random
// directions off neighbouring surface points) cross anything between 0 and ~70 cells.  Here a tile is 1024 rays on 512 lanes:
// a counting sort of the rays by the length of their box crossing (64 bins, LDS atomics; the slab test is the walk's own)
// decides WHICH ray a lane walks in each of its two rounds, so the 64 rays a wave walks together have similar cell counts
// (rays that miss the box share waves that do nothing).  Everything a ray writes goes to its own slot (LDS descriptors, the
// global per-ray records), so outputs -- packed in ray order by the same scan / look-back / element-parallel expansion --
// are bit-identical to traverse_fused_kernel's.  Run descriptors: 4 per ray inline (u16 counts); more runs = the in-order
// re-walk, as before.
constexpr int ST_THREADS = 512, ST_RAYS = 1024, ST_RUNS = 4, ST_BINS = 64;

struct LdsRunSinkST {
    float* tfirst;          // [ST_RUNS][ST_RAYS]
    uint16_t* nsamp;        // [ST_RUNS][ST_RAYS]
    int col;
    int n_runs = 0, run_len = 0, n_samples = 0, n_intervals = 0;
    __device__ __forceinline__ void emit(float t_last, float, bool continuous)
    {
        if (!continuous) {
            if (n_runs > 0 && n_runs <= ST_RUNS) nsamp[(n_runs - 1) * ST_RAYS + col] = (uint16_t)run_len;
            if (n_runs < ST_RUNS) tfirst[n_runs * ST_RAYS + col] = t_last;
            n_runs++;
            run_len = 0;
            n_intervals += 2;
        } else {
            n_intervals++;
        }
        n_samples++;
        run_len++;
    }
    __device__ __forceinline__ void finish()
    {
        if (n_runs > 0 && n_runs <= ST_RUNS) nsamp[(n_runs - 1) * ST_RAYS + col] = (uint16_t)run_len;
    }
};

__global__ __launch_bounds__(ST_THREADS) void traverse_sorted_kernel(
    int64_t n_rays, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const uint32_t* __restrict__ grid_bits, int rx, int ry, int rz, const float* __restrict__ aabb_g,
    const float* __restrict__ near_planes, const float* __restrict__ far_planes, float step_size, float cone_angle,
    uint64_t* __restrict__ tile_state /*[n_tiles] zeroed*/, uint32_t* __restrict__ ticket /*zeroed*/, int n_tiles,
    int64_t cap_edges, int64_t cap_samples, int64_t* __restrict__ totals /*[3]: edges, samples, overflow*/,
    int64_t* __restrict__ iv_pinfo, int64_t* __restrict__ sm_pinfo, float* __restrict__ iv_vals,
    uint8_t* __restrict__ iv_is_left, uint8_t* __restrict__ iv_is_right, int64_t* __restrict__ iv_ray,
    float* __restrict__ sm_vals, int64_t* __restrict__ sm_ray, float* __restrict__ term_planes,
    float* __restrict__ sm_ts /*or NULL*/, float* __restrict__ sm_te /*or NULL*/, float inv_bin)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
    __shared__ float s_tfirst[ST_RUNS * ST_RAYS];
    __shared__ uint16_t s_nsamp[ST_RUNS * ST_RAYS];
    __shared__ uint8_t s_nruns[ST_RAYS];
    __shared__ uint32_t s_cnt[ST_RAYS];                 // edges | samples << 16 of a ray
    __shared__ int s_offE[ST_RAYS + 1], s_offS[ST_RAYS + 1];
    __shared__ uint16_t s_perm[ST_RAYS];
    __shared__ int s_hist[ST_BINS];
    __shared__ unsigned long long s_wave_tot[ST_THREADS / 64];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_tile;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    {
        const int n_words = (rx * ry * rz + 31) >> 5;
        const int n_vec = n_words >> 2;
        const uint4* src = reinterpret_cast<const uint4*>(grid_bits);
        uint4* dst = reinterpret_cast<uint4*>(s_bits);
        for (int i = t; i < n_vec; i += ST_THREADS) dst[i] = src[i];
        for (int i = (n_vec << 2) + t; i < n_words; i += ST_THREADS) s_bits[i] = grid_bits[i];
    }
    float aabb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) aabb[k] = aabb_g[k];
    __shared__ int s_box[6];
    __syncthreads();
    occupied_cell_box(s_bits, rx, ry, rz, s_box, t, ST_THREADS);
    const int* const occ_box = term_planes ? nullptr : s_box;
  for (;;) {
    __syncthreads();                       // previous tile fully written; LDS descriptors free
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
    if (t < ST_BINS) s_hist[t] = 0;
    __syncthreads();
    const int tile = (int)s_tile;
    if (tile >= n_tiles) break;
    const int64_t ray0 = (int64_t)tile * ST_RAYS;
    const int n_here = (int)((n_rays - ray0 < ST_RAYS) ? n_rays - ray0 : ST_RAYS);

    // ---- phase 0: span of every ray's box crossing -> bin; counting sort of the tile's rays by bin
    int bin[2], rank[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = k * ST_THREADS + t;
        bin[k] = 0;
        if (r < n_here) {
            const int64_t g = ray0 + r;
            const float o[3] = {rays_o[g * 3], rays_o[g * 3 + 1], rays_o[g * 3 + 2]};
            const float d[3] = {rays_d[g * 3], rays_d[g * 3 + 1], rays_d[g * 3 + 2]};
            float tmin, tmax;
            if (ray_aabb(o, d, aabb, tmin, tmax)) {
                const float span = fminf(tmax, far_planes[g]) - fmaxf(tmin, near_planes[g]);
                if (span > 0.0f) bin[k] = 1 + min((int)(span * inv_bin), ST_BINS - 2);
            }
        }
        rank[k] = atomicAdd(&s_hist[bin[k]], 1);
    }
    __syncthreads();
    if (wid == 0) {                                    // exclusive scan of the 64 bins
        int v = s_hist[lane], inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off, 64); if (lane >= off) inc += u; }
        s_hist[lane] = inc - v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; k++) s_perm[s_hist[bin[k]] + rank[k]] = (uint16_t)(k * ST_THREADS + t);
    __syncthreads();

    // ---- phase 1: walk; round k, lane t walks the (k * 512 + t)-th ray in span order
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        const int r = s_perm[k * ST_THREADS + t];
        LdsRunSinkST sink{s_tfirst, s_nsamp, r};
        if (r < n_here) {
            const int64_t g = ray0 + r;
            const float o[3] = {rays_o[g * 3], rays_o[g * 3 + 1], rays_o[g * 3 + 2]};
            const float d[3] = {rays_d[g * 3], rays_d[g * 3 + 1], rays_d[g * 3 + 2]};
            const float near_plane = near_planes[g], far_plane = far_planes[g];
            const float t_term = (cone_angle == 0.0f)
                                     ? dda_walk_fast(o, d, aabb, rx, ry, rz, near_plane, far_plane, step_size, s_bits, sink, occ_box)
                                     : dda_walk(o, d, aabb, rx, ry, rz, near_plane, far_plane, step_size, cone_angle, s_bits, sink);
            sink.finish();
            if (term_planes) term_planes[g] = t_term;
        }
        s_nruns[r] = (uint8_t)min(sink.n_runs, 255);
        // the tile keeps per-ray counts in 16 bits: a ray longer than that (a caller's max_extent hint that is not a true bound)
        // raises the overflow flag and the caller falls back to the two-phase protocol, as for a capacity overflow
        if (sink.n_intervals > 0xFFFF || sink.n_samples > 0xFFFF) totals[2] = 1;
        s_cnt[r] = (uint32_t)(sink.n_intervals & 0xFFFF) | ((uint32_t)(sink.n_samples & 0xFFFF) << 16);
    }
    __syncthreads();

    // ---- phase 2: workgroup exclusive scan of the per-ray counts, in ray order (thread t: rays 2t, 2t + 1)
    const uint32_t c0 = s_cnt[2 * t], c1 = s_cnt[2 * t + 1];
    const unsigned long long m0 = (unsigned long long)(c0 & 0xFFFFu) | ((unsigned long long)(c0 >> 16) << 32);
    const unsigned long long m1 = (unsigned long long)(c1 & 0xFFFFu) | ((unsigned long long)(c1 >> 16) << 32);
    const unsigned long long mine = m0 + m1;
    unsigned long long inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long v = __shfl_up(inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave_tot[wid] = inc;
    __syncthreads();
    unsigned long long wave_off = 0, wg_total = 0;
#pragma unroll
    for (int w = 0; w < ST_THREADS / 64; w++) {
        const unsigned long long v = s_wave_tot[w];
        if (w < wid) wave_off += v;
        wg_total += v;
    }
    const unsigned long long excl = wave_off + inc - mine;
    s_offE[2 * t] = (int)(excl & 0xFFFFFFFFull);
    s_offS[2 * t] = (int)(excl >> 32);
    s_offE[2 * t + 1] = (int)((excl + m0) & 0xFFFFFFFFull);
    s_offS[2 * t + 1] = (int)((excl + m0) >> 32);
    if (t == 0) { s_offE[ST_RAYS] = (int)(wg_total & 0xFFFFFFFFull); s_offS[ST_RAYS] = (int)(wg_total >> 32); }
    const int E_wg = (int)(wg_total & 0xFFFFFFFFull), S_wg = (int)(wg_total >> 32);

    // ---- phase 3: decoupled look-back (wave 0); word = flag | samples << 31 | edges
    if (wid == 0) {
        const uint64_t agg = (uint64_t)E_wg | ((uint64_t)S_wg << 31);
        uint64_t exclusive = 0;
        if (tile > 0) {
            if (lane == 0) ts_store(tile_state + tile, TS_AGG | agg);
            int pos = tile - 1;
            for (;;) {
                const int idx = pos - lane;
                uint64_t v = TS_PREFIX;
                if (idx >= 0) {
                    v = ts_load(tile_state + idx);
                    while ((v >> 62) == 0) { __builtin_amdgcn_s_sleep(1); v = ts_load(tile_state + idx); }
                }
                const unsigned long long is_p = __ballot((v >> 62) == 2);
                const int first_p = is_p ? __builtin_ctzll(is_p) : 64;
                uint64_t val = (lane <= first_p) ? (v & TS_MASK) : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
                exclusive += val;
                if (is_p) break;
                pos -= 64;
            }
        }
        if (lane == 0) {
            ts_store(tile_state + tile, TS_PREFIX | (exclusive + agg));
            s_base = exclusive;
            if (tile == n_tiles - 1) {
                const uint64_t tot = exclusive + agg;
                totals[0] = (int64_t)(tot & 0x7FFFFFFFull);
                totals[1] = (int64_t)(tot >> 31);
            }
        }
    }
    __syncthreads();
    const int64_t baseE = (int64_t)(s_base & 0x7FFFFFFFull), baseS = (int64_t)(s_base >> 31);
    if (baseE + E_wg > cap_edges || baseS + S_wg > cap_samples) {
        if (t == 0) totals[2] = 1;
        continue;
    }

    // ---- phase 4: per-ray records, then element-parallel expansion
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = 2 * t + k;
        if (r < n_here) {
            const uint32_t c = s_cnt[r];
            if (iv_pinfo) { iv_pinfo[2 * (ray0 + r)] = baseE + s_offE[r]; iv_pinfo[2 * (ray0 + r) + 1] = (int64_t)(c & 0xFFFFu); }
            if (sm_pinfo) { sm_pinfo[2 * (ray0 + r)] = baseS + s_offS[r]; sm_pinfo[2 * (ray0 + r) + 1] = (int64_t)(c >> 16); }
        }
    }
    for (int j = t; j < E_wg; j += ST_THREADS) {
        int lo = 0, hi = ST_RAYS;
#pragma unroll
        for (int it = 0; it < 10; it++) {
            const int mid = (lo + hi) >> 1;
            if (s_offE[mid] <= j) lo = mid; else hi = mid;
        }
        const int r = lo;
        if (s_nruns[r] > ST_RUNS) continue;                         // written in order by the re-walk below
        int k = j - s_offE[r];
        int sidx = s_offS[r];
        int qn = 0;
        int n = s_nsamp[r];
        while (k > n) { k -= n + 1; sidx += n; qn++; n = s_nsamp[qn * ST_RAYS + r]; }
        float tv = s_tfirst[qn * ST_RAYS + r];
        if (cone_angle == 0.0f) tv = ia_advance(tv, step_size, k);
        else for (int i = 0; i < k; i++) tv = tv + calc_dt(tv, cone_angle, step_size, 1e10f);
        const int64_t ray = ray0 + r;
        const int64_t gi = baseE + j;
        iv_vals[gi] = tv; iv_ray[gi] = ray; iv_is_left[gi] = k < n; iv_is_right[gi] = k > 0;
        if (k < n) {
            const float t_next = tv + calc_dt(tv, cone_angle, step_size, 1e10f);
            const int64_t gs = baseS + sidx + k;
            sm_vals[gs] = (t_next + tv) * 0.5f; sm_ray[gs] = ray;
            if (sm_ts) { sm_ts[gs] = tv; sm_te[gs] = t_next; }
        }
    }
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        const int r = 2 * t + k;
        if (r < n_here && s_nruns[r] > ST_RUNS) {
            const int64_t g = ray0 + r;
            const float o[3] = {rays_o[g * 3], rays_o[g * 3 + 1], rays_o[g * 3 + 2]};
            const float d[3] = {rays_d[g * 3], rays_d[g * 3 + 1], rays_d[g * 3 + 2]};
            DirectWriteSink w{iv_vals, iv_is_left, iv_is_right, iv_ray, sm_vals, sm_ray, baseE + s_offE[r], baseS + s_offS[r], g, sm_ts, sm_te};
            (void)dda_walk(o, d, aabb, rx, ry, rz, near_planes[g], far_planes[g], step_size, cone_angle, s_bits, w);
        }
    }
  }
}

}  // namespace

IA_EXPORT int ia_occgrid_pack_bits(const uint8_t* binaries, int64_t n_cells, uint32_t* bits, ia_stream_t stream)
{
    IA_REQUIRE(n_cells > 0, "n_cells must be > 0");
    pack_bits_kernel<<<ia::cdiv(n_cells, 256), 256, 0, (hipStream_t)stream>>>(binaries, n_cells, bits);
    return ia::check_launch("ia_occgrid_pack_bits");
}

static int check_grid(int rx, int ry, int rz, size_t* lds_bytes)
{
    IA_REQUIRE(rx > 0 && ry > 0 && rz > 0, "grid resolution must be positive");
    const int64_t cells = (int64_t)rx * ry * rz;
    const int64_t bytes = ((cells + 31) / 32) * 4;
    IA_REQUIRE(bytes <= 128 * 1024, "bit-packed grid must fit in 128 KiB of LDS (<= 1M cells)");
    *lds_bytes = (size_t)((bytes + 15) / 16 * 16);
    return IA_OK;
}

IA_EXPORT int64_t ia_traverse_scratch_bytes(int64_t n_rays) { return (int64_t)sizeof(RayScratch) * (n_rays > 0 ? n_rays : 1); }

static void set_attrs()
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)traverse_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void*)traverse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipGetLastError();
        attr_set = true;
    }
}

IA_EXPORT int ia_traverse_grids_count(int64_t n_rays, const float* rays_o, const float* rays_d,
                                      const uint32_t* grid_bits, int rx, int ry, int rz, const float* aabb,
                                      const float* near_planes, const float* far_planes, float step_size,
                                      float cone_angle, void* scratch, int64_t* packed_counts, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    size_t lds;
    int r = check_grid(rx, ry, rz, &lds);
    if (r != IA_OK) return r;
    IA_REQUIRE(step_size > 0.0f, "step_size must be > 0 (the render_step path always marches with a positive step)");
    set_attrs();
    traverse_kernel<true><<<ia::cdiv(n_rays, TR_THREADS), TR_THREADS, lds, (hipStream_t)stream>>>(
        n_rays, rays_o, rays_d, grid_bits, rx, ry, rz, aabb, near_planes, far_planes, step_size, cone_angle,
        (RayScratch*)scratch, packed_counts, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    return ia::check_launch("ia_traverse_grids_count");
}

IA_EXPORT int ia_traverse_grids_fill(int64_t n_rays, const float* rays_o, const float* rays_d,
                                     const uint32_t* grid_bits, int rx, int ry, int rz, const float* aabb,
                                     const float* near_planes, const float* far_planes, float step_size,
                                     float cone_angle, const void* scratch, const int64_t* packed_counts,
                                     const int64_t* packed_starts, int64_t* iv_packed_info, int64_t* sm_packed_info,
                                     float* iv_vals, uint8_t* iv_is_left, uint8_t* iv_is_right,
                                     int64_t* iv_ray_indices, float* sm_vals, int64_t* sm_ray_indices,
                                     float* termination_planes, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    size_t lds;
    int r = check_grid(rx, ry, rz, &lds);
    if (r != IA_OK) return r;
    set_attrs();
    const int grid = ia::cdiv(n_rays, TR_THREADS);
    expand_kernel<<<grid, TR_THREADS, 0, (hipStream_t)stream>>>(
        n_rays, (const RayScratch*)scratch, packed_counts, packed_starts, step_size, cone_angle, iv_packed_info,
        sm_packed_info, iv_vals, iv_is_left, iv_is_right, iv_ray_indices, sm_vals, sm_ray_indices, termination_planes);
    // rays with more than TR_RUNS runs: workgroups without any such ray exit before staging the grid
    traverse_kernel<false><<<grid, TR_THREADS, lds, (hipStream_t)stream>>>(
        n_rays, rays_o, rays_d, grid_bits, rx, ry, rz, aabb, near_planes, far_planes, step_size, cone_angle,
        (RayScratch*)scratch, nullptr, true, packed_starts, iv_vals, iv_is_left, iv_is_right, iv_ray_indices, sm_vals,
        sm_ray_indices, nullptr);
    return ia::check_launch("ia_traverse_grids_fill");
}

IA_EXPORT int64_t ia_traverse_fused_scratch_bytes(int64_t n_rays)
{
    const int64_t tiles = (n_rays + TR_THREADS - 1) / TR_THREADS;
    return (tiles + 2) * 8 + 64;
}

IA_EXPORT int ia_traverse_grids_fused(int64_t n_rays, const float* rays_o, const float* rays_d,
                                      const uint32_t* grid_bits, int rx, int ry, int rz, const float* aabb,
                                      const float* near_planes, const float* far_planes, float step_size,
                                      float cone_angle, void* scratch, int64_t cap_edges, int64_t cap_samples,
                                      int64_t* totals, int64_t* iv_packed_info, int64_t* sm_packed_info, float* iv_vals,
                                      uint8_t* iv_is_left, uint8_t* iv_is_right, int64_t* iv_ray_indices, float* sm_vals,
                                      int64_t* sm_ray_indices, float* termination_planes, float* sm_t_starts,
                                      float* sm_t_ends, int span_sorted, ia_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    IA_REQUIRE((sm_t_starts == nullptr) == (sm_t_ends == nullptr), "sm_t_starts and sm_t_ends come together");
    ia::zero_bytes(totals, 3 * sizeof(int64_t), s);
    if (n_rays == 0) return IA_OK;
    size_t lds;
    int r = check_grid(rx, ry, rz, &lds);
    if (r != IA_OK) return r;
    IA_REQUIRE(step_size > 0.0f, "step_size must be > 0");
    IA_REQUIRE(cap_edges < (1ll << 31) && cap_samples < (1ll << 31), "capacities must be < 2^31 (split the ray batch)");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)traverse_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipGetLastError();
        attr_set = true;
    }
    const int64_t sb = ia_traverse_fused_scratch_bytes(n_rays);
    ia::zero_bytes(scratch, ((size_t)sb + 3) & ~(size_t)3, s);
    // span-sorted tiles (1024 rays on 512 lanes): for INCOHERENT batches -- the secondary march: 10.7 -> 8.6 ms per headline step.
    // Coherent primary rays and dense grids are faster in ray order (85 vs 95 us per 540x540 frame; 238 vs 290 us on a dense
    // grid, where the walk is short and the expansion dominates), so the caller chooses; env IA_TRAVERSE_TILES = ray | span
    // overrides for A / B runs.  Per-ray counts are kept in 16 bits there.
    bool sorted = span_sorted != 0 && n_rays >= (1 << 14) && lds <= 32 * 1024 + 64;
    if (const char* tv = getenv("IA_TRAVERSE_TILES")) sorted = (tv[0] == 's') && lds <= 32 * 1024 + 64;
    if (sorted) {
        // the longest crossing: span_sorted > 1 is the caller's bound of the samples per ray (capacities may be sized for the AVERAGE ray),
        // else what the capacities allow, cap_samples / n_rays steps; bins of 1/62 of that
        const double max_steps = span_sorted > 1 ? (double)span_sorted : (double)cap_samples / (double)n_rays;
        if (max_steps < 30000.0) {
            static bool attr2 = false;
            if (!attr2) {
                (void)hipFuncSetAttribute((const void*)traverse_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                (void)hipGetLastError();
                attr2 = true;
            }
            const int tiles_s = ia::cdiv(n_rays, ST_RAYS);
            uint64_t* state_s = (uint64_t*)scratch;
            uint32_t* ticket_s = (uint32_t*)(state_s + tiles_s + 1);
            const float inv_bin = (float)(62.0 / (max_steps * (double)step_size));
            const int grid_s = tiles_s < 512 ? tiles_s : 512;      // 2 resident workgroups per CU
            traverse_sorted_kernel<<<grid_s, ST_THREADS, lds, s>>>(
                n_rays, rays_o, rays_d, grid_bits, rx, ry, rz, aabb, near_planes, far_planes, step_size, cone_angle, state_s, ticket_s,
                tiles_s, cap_edges, cap_samples, totals, iv_packed_info, sm_packed_info, iv_vals, iv_is_left, iv_is_right,
                iv_ray_indices, sm_vals, sm_ray_indices, termination_planes, sm_t_starts, sm_t_ends, inv_bin);
            return ia::check_launch("ia_traverse_grids_fused");
        }
    }
    const int tiles = ia::cdiv(n_rays, TR_THREADS);
    uint64_t* state = (uint64_t*)scratch;
    uint32_t* ticket = (uint32_t*)(state + tiles + 1);
    const int grid = tiles < 768 ? tiles : 768;        // 3 resident workgroups per CU (53 KB of LDS each)
    traverse_fused_kernel<<<grid, TR_THREADS, lds, s>>>(
        n_rays, rays_o, rays_d, grid_bits, rx, ry, rz, aabb, near_planes, far_planes, step_size, cone_angle, state, ticket,
        tiles, cap_edges, cap_samples, totals, iv_packed_info, sm_packed_info, iv_vals, iv_is_left, iv_is_right,
        iv_ray_indices, sm_vals, sm_ray_indices, termination_planes, sm_t_starts, sm_t_ends);
    return ia::check_launch("ia_traverse_grids_fused");
}
