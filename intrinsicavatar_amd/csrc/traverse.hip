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
#include "ia_common.h"

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
