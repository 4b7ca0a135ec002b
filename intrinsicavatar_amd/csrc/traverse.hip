// traverse.hip -- occupancy-grid ray marching for gfx950 (replaces nerfacc.traverse_grids;
// reference call sites models/occ_grid/temporal_occ_grid.py:166-175, models/intrinsic_avatar.py:84-93).
//
// Layout / mapping
//   * the 64^3 occupancy grid is bit-packed (32 KiB) and staged ONCE per workgroup into LDS
//     with coalesced 16-byte loads; every cell test of the DDA is then a ds_read_b32 + bit test;
//   * one ray per lane, 256 rays per workgroup; rays of one workgroup are pixel-neighbours, so
//     a wave's lanes walk neighbouring cells (LDS broadcast / few bank conflicts);
//   * two passes (count, fill) with an exclusive scan in between -- output order is
//     ray order then marching order, exactly as upstream.
// Arithmetic is kept operation-for-operation identical to oracle/ia_oracle.c (this TU is
// built with -ffp-contract=off) so edge/sample counts and t values are bit-exact.
#include "ia_common.h"

namespace {

constexpr int TR_THREADS = 256;

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
    int64_t* __restrict__ iv_cnt, int64_t* __restrict__ sm_cnt,
    const int64_t* __restrict__ iv_start, const int64_t* __restrict__ sm_start,
    float* __restrict__ iv_vals, uint8_t* __restrict__ iv_is_left, uint8_t* __restrict__ iv_is_right,
    int64_t* __restrict__ iv_ray, float* __restrict__ sm_vals, int64_t* __restrict__ sm_ray,
    float* __restrict__ term_planes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
    const int n_words = (rx * ry * rz + 31) >> 5;
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
    if (!FIRST_PASS) { iv_base = iv_start[tid]; sm_base = sm_start[tid]; }
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
    if (FIRST_PASS) { iv_cnt[tid] = n_intervals; sm_cnt[tid] = n_samples; }
    else if (term_planes) term_planes[tid] = t_last;
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
    IA_REQUIRE(bytes <= 160 * 1024, "bit-packed grid must fit the 160 KiB LDS (<= 1.3M cells)");
    *lds_bytes = (size_t)((bytes + 15) / 16 * 16);
    return IA_OK;
}

IA_EXPORT int ia_traverse_grids_count(int64_t n_rays, const float* rays_o, const float* rays_d,
                                      const uint32_t* grid_bits, int rx, int ry, int rz, const float* aabb,
                                      const float* near_planes, const float* far_planes, float step_size,
                                      float cone_angle, int64_t* iv_cnt, int64_t* sm_cnt, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    size_t lds;
    int r = check_grid(rx, ry, rz, &lds);
    if (r != IA_OK) return r;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)traverse_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)traverse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    traverse_kernel<true><<<ia::cdiv(n_rays, TR_THREADS), TR_THREADS, lds, (hipStream_t)stream>>>(
        n_rays, rays_o, rays_d, grid_bits, rx, ry, rz, aabb, near_planes, far_planes, step_size, cone_angle, iv_cnt,
        sm_cnt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    return ia::check_launch("ia_traverse_grids_count");
}

IA_EXPORT int ia_traverse_grids_fill(int64_t n_rays, const float* rays_o, const float* rays_d,
                                     const uint32_t* grid_bits, int rx, int ry, int rz, const float* aabb,
                                     const float* near_planes, const float* far_planes, float step_size,
                                     float cone_angle, const int64_t* iv_start, const int64_t* sm_start,
                                     float* iv_vals, uint8_t* iv_is_left, uint8_t* iv_is_right,
                                     int64_t* iv_ray_indices, float* sm_vals, int64_t* sm_ray_indices,
                                     float* termination_planes, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    size_t lds;
    int r = check_grid(rx, ry, rz, &lds);
    if (r != IA_OK) return r;
    (void)hipFuncSetAttribute((const void*)traverse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    traverse_kernel<false><<<ia::cdiv(n_rays, TR_THREADS), TR_THREADS, lds, (hipStream_t)stream>>>(
        n_rays, rays_o, rays_d, grid_bits, rx, ry, rz, aabb, near_planes, far_planes, step_size, cone_angle, nullptr,
        nullptr, iv_start, sm_start, iv_vals, iv_is_left, iv_is_right, iv_ray_indices, sm_vals, sm_ray_indices,
        termination_planes);
    return ia::check_launch("ia_traverse_grids_fill");
}
