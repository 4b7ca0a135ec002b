// volint.hip -- the host logic between K1 (ray_resampling) and the PBR estimator as kernels for gfx950.
//
// Replaces the torch op sequence of
//   sample_volume_interaction           models/pbr/utils.py:130-229   (2 x nonzero, unpack_info, 9 advanced-index gathers,
//                                                                      2 x scatter_ of the re-sampled weights)
//   the light-direction shuffle         models/intrinsic_avatar.py:1356-1378 (CPU rand + argsort + unpack_data + pack_data)
//   Lo.scatter_ + accumulate_along_rays models/intrinsic_avatar.py:1335-1342,1420-1466
//   emitter.sample                      lib.torch_pbr (call sites :300-305,:772-776)
//   boolean-mask compaction of the secondary rays   :788-803
//
// Structure of K1's output that makes all of this scan + streaming work (cdf.cu:46-148): every ray with samples owns
// exactly `spp` consecutive re-samples; its foreground re-samples come first (j < spp - bg_count[ray]), in non-decreasing
// order of the sampled interval, the background tail after them.  Hence
//   * the foreground list is [ray-major, j] with per-ray start = exclusive scan of (spp - bg_count) -- no nonzero();
//   * the foreground re-samples of one source interval s are a CONTIGUOUS range of that list, starting at the exclusive
//     scan of fg_counts[s] -- the backward of the attribute gathers is a segmented sum without atomics;
//   * the composite is a per-ray sum over a contiguous range plus transmittance x background colour.
// Floating point: sums over a ray run lane-strided + wave tree (deterministic; the reference's index_add_ is an
// unordered atomic sum), everything else is elementwise.
#include "ia_common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- per-ray foreground counts ------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void vi_layout_kernel(int64_t n_rays, int spp, const int32_t* __restrict__ rpi,
                                                             const int32_t* __restrict__ bg_cnt, int32_t* __restrict__ fg_ray_cnt)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    fg_ray_cnt[r] = rpi[2 * r + 1] > 0 ? spp - bg_cnt[r] : 0;
}

// ---- gather: one workgroup per ray ----------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void vi_gather_kernel(
    int64_t n_rays, const int32_t* __restrict__ rpi, const int32_t* __restrict__ fg_ray_cnt, const int32_t* __restrict__ fg_start,
    const float* __restrict__ ts, const int64_t* __restrict__ sidx, const int32_t* __restrict__ fg_cnt,
    const float* __restrict__ weights, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ normals, const float* __restrict__ albedo, const float* __restrict__ rough,
    const float* __restrict__ metal, int32_t* __restrict__ fg_src, int32_t* __restrict__ fg_ray, float* __restrict__ pos,
    float* __restrict__ view, float* __restrict__ o_nrm, float* __restrict__ o_alb, float* __restrict__ o_rough,
    float* __restrict__ o_metal, float* __restrict__ o_rw)
{
    const int64_t r = blockIdx.x;
    const int nf = fg_ray_cnt[r];
    if (nf == 0) return;
    const int64_t base = rpi[2 * r], fs = fg_start[r];
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    for (int j = threadIdx.x; j < nf; j += THREADS) {
        const int64_t i = base + j, k = fs + j;
        const int64_t s = sidx[i];
        const float t = ts[i];
        fg_src[k] = (int32_t)s;
        fg_ray[k] = (int32_t)r;
        pos[3 * k] = ox + dx * t; pos[3 * k + 1] = oy + dy * t; pos[3 * k + 2] = oz + dz * t;
        view[3 * k] = dx; view[3 * k + 1] = dy; view[3 * k + 2] = dz;
        o_nrm[3 * k] = normals[3 * s]; o_nrm[3 * k + 1] = normals[3 * s + 1]; o_nrm[3 * k + 2] = normals[3 * s + 2];
        o_alb[3 * k] = albedo[3 * s]; o_alb[3 * k + 1] = albedo[3 * s + 1]; o_alb[3 * k + 2] = albedo[3 * s + 2];
        o_rough[k] = rough[s];
        o_metal[k] = metal[s];
        o_rw[k] = weights[s] / (float)fg_cnt[s];
    }
}

// ---- the reference's index lists (drop-in form of sample_volume_interaction): fg / bg indices, ray index and the
// re-sampled weight of every re-sample, one workgroup per ray ------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void vi_indices_kernel(
    int64_t n_rays, int spp, const int32_t* __restrict__ rpi, const int32_t* __restrict__ fg_ray_cnt,
    const int32_t* __restrict__ fg_start, const int32_t* __restrict__ bg_cnt, const int64_t* __restrict__ sidx,
    const int32_t* __restrict__ fg_cnt, const float* __restrict__ weights, const float* __restrict__ transmittance,
    int64_t* __restrict__ fg_indices, int64_t* __restrict__ bg_indices, int64_t* __restrict__ ray_indices, float* __restrict__ rw)
{
    const int64_t r = blockIdx.x;
    const int cnt = rpi[2 * r + 1];
    if (cnt <= 0) return;
    const int64_t base = rpi[2 * r];
    const int nf = fg_ray_cnt[r];
    const int64_t fs = fg_start[r], bs = base - fs;          // bg list position = re-samples before this ray - fg before it
    const float wb = (spp - nf) > 0 ? transmittance[r] / (float)bg_cnt[r] : 0.0f;
    for (int j = threadIdx.x; j < cnt; j += THREADS) {
        const int64_t i = base + j;
        if (ray_indices) ray_indices[i] = r;
        if (j < nf) {
            if (fg_indices) fg_indices[fs + j] = i;
            const int64_t s = sidx[i];
            if (rw) rw[i] = weights[s] / (float)fg_cnt[s];
        } else {
            if (bg_indices) bg_indices[bs + (j - nf)] = i;
            if (rw) rw[i] = wb;
        }
    }
}

// ---- gather backward: segmented sums, one wave per group of 64 source intervals -------------------------------------
__global__ __launch_bounds__(THREADS) void vi_gather_bwd_kernel(
    int64_t S, const int32_t* __restrict__ fg_cnt, const int32_t* __restrict__ fg_off, const float* __restrict__ g_nrm,
    const float* __restrict__ g_alb, const float* __restrict__ g_rough, const float* __restrict__ g_metal,
    const float* __restrict__ g_rw, float* __restrict__ g_normals, float* __restrict__ g_albedo, float* __restrict__ g_r,
    float* __restrict__ g_m, float* __restrict__ g_weights)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * THREADS + threadIdx.x) >> 6;
    const int64_t s0 = wave * 64;
    if (s0 >= S) return;
    const int64_t s_mine = s0 + lane;
    const int c_mine = s_mine < S ? fg_cnt[s_mine] : 0;
    const int o_mine = s_mine < S ? fg_off[s_mine] : 0;
    float out[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};       // this lane's interval: g_normal(3) g_albedo(3) g_rough g_metal g_weight
    unsigned long long todo = __ballot(c_mine > 0);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int c = __shfl(c_mine, l, 64);
        const int64_t o = __shfl(o_mine, l, 64);
        float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = lane; j < c; j += 64) {
            const int64_t k = o + j;
            if (g_nrm) { a[0] += g_nrm[3 * k]; a[1] += g_nrm[3 * k + 1]; a[2] += g_nrm[3 * k + 2]; }
            if (g_alb) { a[3] += g_alb[3 * k]; a[4] += g_alb[3 * k + 1]; a[5] += g_alb[3 * k + 2]; }
            if (g_rough) a[6] += g_rough[k];
            if (g_metal) a[7] += g_metal[k];
            if (g_rw) a[8] += g_rw[k];
        }
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const float t = wave_sum(a[q]);
            const float tt = __shfl(t, 0, 64);
            if (lane == l) out[q] = tt;
        }
    }
    if (s_mine < S) {
        g_normals[3 * s_mine] = out[0]; g_normals[3 * s_mine + 1] = out[1]; g_normals[3 * s_mine + 2] = out[2];
        g_albedo[3 * s_mine] = out[3]; g_albedo[3 * s_mine + 1] = out[4]; g_albedo[3 * s_mine + 2] = out[5];
        g_r[s_mine] = out[6];
        g_m[s_mine] = out[7];
        g_weights[s_mine] = c_mine > 0 ? out[8] / (float)c_mine : 0.0f;
    }
}

// ---- composite: rgb[r] = sum_k rw[k] Lo[k] + [bg_cnt > 0] T[r] bg ; rays without samples: bg ------------------------
__global__ __launch_bounds__(THREADS) void vi_composite_kernel(
    int64_t n_rays, const int32_t* __restrict__ rpi, const int32_t* __restrict__ fg_ray_cnt, const int32_t* __restrict__ fg_start,
    const int32_t* __restrict__ bg_cnt, const float* __restrict__ rw, const float* __restrict__ Lo,
    const float* __restrict__ transmittance, const float* __restrict__ bg, const float* __restrict__ bg_rays /*[n,3] or NULL*/,
    float* __restrict__ rgb)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * THREADS + threadIdx.x) >> 6;      // one wave per ray
    if (r >= n_rays) return;
    const float b0 = bg_rays ? bg_rays[3 * r] : bg[0], b1 = bg_rays ? bg_rays[3 * r + 1] : bg[1],
                b2 = bg_rays ? bg_rays[3 * r + 2] : bg[2];
    if (rpi[2 * r + 1] <= 0) {
        if (lane == 0) { rgb[3 * r] = b0; rgb[3 * r + 1] = b1; rgb[3 * r + 2] = b2; }
        return;
    }
    const int nf = fg_ray_cnt[r];
    const int64_t fs = fg_start[r];
    float a0 = 0, a1 = 0, a2 = 0;
    for (int j = lane; j < nf; j += 64) {
        const int64_t k = fs + j;
        const float w = rw[k];
        a0 += w * Lo[3 * k]; a1 += w * Lo[3 * k + 1]; a2 += w * Lo[3 * k + 2];
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) {
        const float T = bg_cnt[r] > 0 ? transmittance[r] : 0.0f;
        rgb[3 * r] = a0 + T * b0; rgb[3 * r + 1] = a1 + T * b1; rgb[3 * r + 2] = a2 + T * b2;
    }
}

__global__ __launch_bounds__(THREADS) void vi_composite_bwd_kernel(
    int64_t F, const int32_t* __restrict__ fg_ray, const float* __restrict__ rw, const float* __restrict__ Lo,
    const float* __restrict__ g_rgb, float* __restrict__ g_rw, float* __restrict__ g_Lo)
{
    const int64_t k = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (k >= F) return;
    const int64_t r = fg_ray[k];
    const float g0 = g_rgb[3 * r], g1 = g_rgb[3 * r + 1], g2 = g_rgb[3 * r + 2];
    const float w = rw[k];
    if (g_rw) g_rw[k] = Lo[3 * k] * g0 + Lo[3 * k + 1] * g1 + Lo[3 * k + 2] * g2;
    if (g_Lo) { g_Lo[3 * k] = w * g0; g_Lo[3 * k + 1] = w * g1; g_Lo[3 * k + 2] = w * g2; }
}

__global__ __launch_bounds__(THREADS) void vi_composite_bwd_T_kernel(int64_t n_rays, const int32_t* __restrict__ rpi,
                                                                      const int32_t* __restrict__ bg_cnt,
                                                                      const float* __restrict__ bg,
                                                                      const float* __restrict__ bg_rays /*[n,3] or NULL*/,
                                                                      const float* __restrict__ g_rgb, float* __restrict__ g_T)
{
    const int64_t r = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (r >= n_rays) return;
    const bool has = rpi[2 * r + 1] > 0 && bg_cnt[r] > 0;
    const float b0 = bg_rays ? bg_rays[3 * r] : bg[0], b1 = bg_rays ? bg_rays[3 * r + 1] : bg[1],
                b2 = bg_rays ? bg_rays[3 * r + 2] : bg[2];                  // the colour the forward multiplied T with
    g_T[r] = has ? b0 * g_rgb[3 * r] + b1 * g_rgb[3 * r + 1] + b2 * g_rgb[3 * r + 2] : 0.0f;
}

// ---- per-ray permutation of [0, spp): argsort of uniforms, ties by index (stable) -----------------------------------
// one workgroup per ray; bitonic sort of 64-bit keys (float bits << 32 | index) in LDS; uniforms are in [0, 1) so the
// IEEE bit pattern orders like the value.  Only the first fg_ray_cnt[r] entries of the permutation are consumed.
__global__ __launch_bounds__(THREADS) void light_shuffle_kernel(int64_t n_rays, int spp, int npow2,
                                                                 const int32_t* __restrict__ fg_ray_cnt,
                                                                 const int32_t* __restrict__ fg_start, const float* __restrict__ u,
                                                                 int32_t* __restrict__ shuffled)
{
    extern __shared__ unsigned long long s_key[];
    const int64_t r = blockIdx.x;
    const int nf = fg_ray_cnt[r];
    if (nf == 0) return;
    for (int i = threadIdx.x; i < npow2; i += THREADS) {
        unsigned long long k = ~0ull;
        if (i < spp) {
            float v = u[r * spp + i];
            v = v + 0.0f;                                               // -0.0 -> +0.0
            k = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)i;
        }
        s_key[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (npow2 >> 1); t += THREADS) {
                const int lo = (t / stride) * (stride << 1) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = s_key[lo], b = s_key[hi];
                if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
            }
            __syncthreads();
        }
    }
    const int64_t fs = fg_start[r];
    for (int j = threadIdx.x; j < nf; j += THREADS) shuffled[fs + j] = (int32_t)(s_key[j] & 0xffffffffu);
}

// ---- emitter.sample: inverse CDF over the flattened pmf + uniform jitter inside the texel ----------------------------
constexpr int ES_PER_WG = 4096;            // samples per workgroup (the block table is loaded once per workgroup)
__global__ __launch_bounds__(THREADS) void envlight_sample_kernel(int64_t k, const float* __restrict__ u, const double* __restrict__ cdf,
                                                                   int H, int W, int log2b, const float* __restrict__ rot /*[9] or NULL*/,
                                                                   float* __restrict__ dirs)
{
    // searchsorted(cdf, target, right=True) -- the first index with cdf[idx] > target -- in two levels: the LAST entry of every block of
    // 2^log2b entries sits in LDS (<= 4096 doubles, loaded once per workgroup of ES_PER_WG samples); the first block whose last entry is
    // above the target holds the answer (cdf is non-decreasing), and the search inside it touches one or two cache lines instead of the
    // ~20 dependent loads of a search over the whole map.  Same index as the one-level search.
    extern __shared__ __attribute__((aligned(16))) double s_last[];
    const int64_t n = (int64_t)H * W;
    const int64_t bsz = (int64_t)1 << log2b;
    const int nb = (int)((n + bsz - 1) >> log2b);
    for (int j = threadIdx.x; j < nb; j += THREADS) { const int64_t e = ((int64_t)(j + 1) << log2b); s_last[j] = cdf[(e < n ? e : n) - 1]; }
    __syncthreads();
    const double total = s_last[nb - 1];
  for (int64_t i = (int64_t)blockIdx.x * ES_PER_WG + threadIdx.x, i_end = min(k, ((int64_t)blockIdx.x + 1) * ES_PER_WG); i < i_end; i += THREADS) {
    const double target = (double)u[3 * i] * total;
    int b0 = 0, b1 = nb;
    while (b0 < b1) {
        const int mid = (b0 + b1) >> 1;
        if (s_last[mid] > target) b1 = mid; else b0 = mid + 1;
    }
    int64_t lo = n;
    if (b0 < nb) {
        int64_t hi = min(((int64_t)(b0 + 1)) << log2b, n);
        lo = (int64_t)b0 << log2b;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] > target) hi = mid; else lo = mid + 1;
        }
    }
    if (lo > n - 1) lo = n - 1;
    const int64_t y = lo / W, x = lo % W;
    const double uu = ((double)x + (double)u[3 * i + 1]) / (double)W, vv = ((double)y + (double)u[3 * i + 2]) / (double)H;
    const double pi = 3.14159265358979323846;
    const double phi = (uu - 0.5) * 2.0 * pi, th = vv * pi;
    const double st = sin(th);
    float d0 = (float)(st * sin(phi)), d1 = (float)cos(th), d2 = (float)(-st * cos(phi));
    if (rot) {
        // transform_dirs_w2s (snarf_deformer.py:149-155): F.normalize(d @ R^T, eps = 1e-6)
        const float e0 = d0 * rot[0] + d1 * rot[1] + d2 * rot[2];
        const float e1 = d0 * rot[3] + d1 * rot[4] + d2 * rot[5];
        const float e2 = d0 * rot[6] + d1 * rot[7] + d2 * rot[8];
        const float nn = fmaxf(sqrtf(e0 * e0 + e1 * e1 + e2 * e2), 1e-6f);
        d0 = e0 / nn; d1 = e1 / nn; d2 = e2 / nn;
    }
    dirs[3 * i] = d0; dirs[3 * i + 1] = d1; dirs[3 * i + 2] = d2;
  }
}

// ---- secondary rays: cosine mask -> compact list ----------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void cos_mask_kernel(int64_t F, const float* __restrict__ nrm, const float* __restrict__ dirs,
                                                            const int32_t* __restrict__ dir_index /*[F] or NULL*/,
                                                            int32_t* __restrict__ flag)
{
    const int64_t k = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (k >= F) return;
    const int64_t q = dir_index ? dir_index[k] : k;
    const float c = nrm[3 * k] * dirs[3 * q] + nrm[3 * k + 1] * dirs[3 * q + 1] + nrm[3 * k + 2] * dirs[3 * q + 2];
    flag[k] = c > 1e-6f ? 1 : 0;
}

__global__ __launch_bounds__(THREADS) void compact_rays_kernel(int64_t F, const int32_t* __restrict__ flag,
                                                                const int32_t* __restrict__ slot, const float* __restrict__ pos,
                                                                const float* __restrict__ dirs, const int32_t* __restrict__ dir_index,
                                                                float* __restrict__ ro, float* __restrict__ rd,
                                                                int32_t* __restrict__ src, float* __restrict__ dense_dirs)
{
    const int64_t k = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (k >= F) return;
    const int64_t q = dir_index ? dir_index[k] : k;
    const float d0 = dirs[3 * q], d1 = dirs[3 * q + 1], d2 = dirs[3 * q + 2];
    if (dense_dirs) { dense_dirs[3 * k] = d0; dense_dirs[3 * k + 1] = d1; dense_dirs[3 * k + 2] = d2; }
    if (!flag[k]) return;
    const int64_t m = slot[k];
    ro[3 * m] = pos[3 * k]; ro[3 * m + 1] = pos[3 * k + 1]; ro[3 * m + 2] = pos[3 * k + 2];
    rd[3 * m] = d0; rd[3 * m + 1] = d1; rd[3 * m + 2] = d2;
    src[m] = (int32_t)k;
}

__global__ __launch_bounds__(THREADS) void scatter_secondary_kernel(int64_t M, const int32_t* __restrict__ src,
                                                                     const float* __restrict__ tr, const float* __restrict__ rgb,
                                                                     float* __restrict__ dense_tr, float* __restrict__ dense_rgb)
{
    const int64_t m = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (m >= M) return;
    const int64_t k = src[m];
    dense_tr[k] = fminf(fmaxf(tr[m], 0.0f), 1.0f);                     // secondary_tr.clamp_(0, 1), :803
    dense_rgb[3 * k] = rgb[3 * m]; dense_rgb[3 * k + 1] = rgb[3 * m + 1]; dense_rgb[3 * k + 2] = rgb[3 * m + 2];
}

// the same result written point by point (every dense element exactly once: no zero fill of [F,4] beforehand -- 1.3 GB per headline step):
// point k reads its ray's results at slot[k] when flag[k] is set, zeros otherwise
__global__ __launch_bounds__(THREADS) void gather_secondary_kernel(int64_t F, const int32_t* __restrict__ flag, const int32_t* __restrict__ slot,
                                                                    const float* __restrict__ tr, const float* __restrict__ rgb,
                                                                    float* __restrict__ dense_tr, float* __restrict__ dense_rgb)
{
    const int64_t k = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (k >= F) return;
    float t = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    if (flag[k]) {
        const int64_t m = slot[k];
        t = fminf(fmaxf(tr[m], 0.0f), 1.0f);                           // secondary_tr.clamp_(0, 1), :803
        r = rgb[3 * m]; g = rgb[3 * m + 1]; b = rgb[3 * m + 2];
    }
    dense_tr[k] = t;
    dense_rgb[3 * k] = r; dense_rgb[3 * k + 1] = g; dense_rgb[3 * k + 2] = b;
}

// ---- spatial ordering of query points: 30-bit Morton code of the cell (origin, 1 / cell size), for a key-value sort ----
__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    return (v | (v << 2)) & 0x09249249u;
}

__global__ __launch_bounds__(THREADS) void morton_keys_kernel(int64_t n, const float* __restrict__ pts, float ox, float oy, float oz,
                                                               float inv_cell, int32_t* __restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float fx = (pts[3 * i] - ox) * inv_cell, fy = (pts[3 * i + 1] - oy) * inv_cell, fz = (pts[3 * i + 2] - oz) * inv_cell;
    const uint32_t x = (uint32_t)fminf(fmaxf(fx, 0.0f), 1023.0f), y = (uint32_t)fminf(fmaxf(fy, 0.0f), 1023.0f),
                   z = (uint32_t)fminf(fmaxf(fz, 0.0f), 1023.0f);
    keys[i] = (int32_t)(spread10(x) | (spread10(y) << 1) | (spread10(z) << 2));
}

__global__ __launch_bounds__(THREADS) void gather_rows3_kernel(int64_t n, const float* __restrict__ src, const int64_t* __restrict__ order,
                                                                float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t j = order[i];
    dst[3 * i] = src[3 * j]; dst[3 * i + 1] = src[3 * j + 1]; dst[3 * i + 2] = src[3 * j + 2];
}

__global__ __launch_bounds__(THREADS) void scatter_f32_kernel(int64_t n, const float* __restrict__ src, const int64_t* __restrict__ order,
                                                               float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    dst[order[i]] = src[i];
}

// ---- GaussianHistogram (models/utils.py:133-149), the soft histogram of the albedo-entropy regulariser --------------------
//   h_b = sum_n exp(-0.5 ((x_n - c_b) / sigma)^2) / (sigma sqrt(2 pi)) * delta ,  c_b = min + delta (b + 0.5)
constexpr int GH_MAX_BINS = 32;

__global__ __launch_bounds__(THREADS) void gauss_hist_kernel(int64_t n, const float* __restrict__ x, const float* __restrict__ sigma_p,
                                                              int bins, float vmin, float delta, float* __restrict__ out)
{
    __shared__ float s_acc[GH_MAX_BINS];
    if (threadIdx.x < GH_MAX_BINS) s_acc[threadIdx.x] = 0.0f;
    __syncthreads();
    const float sigma = *sigma_p;
    const float norm = delta / (sigma * 2.5066282746310002f);
    float acc[GH_MAX_BINS];
#pragma unroll
    for (int b = 0; b < GH_MAX_BINS; b++) acc[b] = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float xi = x[i];
#pragma unroll
        for (int b = 0; b < GH_MAX_BINS; b++) {
            if (b < bins) {
                const float z = (xi - (vmin + delta * ((float)b + 0.5f))) / sigma;
                acc[b] += expf(-0.5f * z * z) * norm;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < GH_MAX_BINS; b++) {
        if (b < bins) {
            const float t = wave_sum(acc[b]);
            if ((threadIdx.x & 63) == 0) atomicAdd(&s_acc[b], t);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < bins) atomicAdd(&out[threadIdx.x], s_acc[threadIdx.x]);
}

// g_x[n] = sum_b g_b dh_nb/dx ;  g_sigma += sum_n sum_b g_b dh_nb/dsigma
__global__ __launch_bounds__(THREADS) void gauss_hist_bwd_kernel(int64_t n, const float* __restrict__ x, const float* __restrict__ sigma_p,
                                                                  int bins, float vmin, float delta, const float* __restrict__ g_out,
                                                                  float* __restrict__ g_x, float* __restrict__ g_sigma)
{
    __shared__ float s_gs[THREADS / 64];
    const float sigma = *sigma_p;
    const float norm = delta / (sigma * 2.5066282746310002f);
    float gs = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float xi = x[i];
        float gx = 0.0f;
        for (int b = 0; b < bins; b++) {
            const float d = xi - (vmin + delta * ((float)b + 0.5f));
            const float z = d / sigma;
            const float h = expf(-0.5f * z * z) * norm * g_out[b];
            gx += h * (-d / (sigma * sigma));
            gs += h * (z * z - 1.0f) / sigma;
        }
        g_x[i] = gx;
    }
    gs = wave_sum(gs);
    if ((threadIdx.x & 63) == 0) s_gs[threadIdx.x >> 6] = gs;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < THREADS / 64; w++) t += s_gs[w];
        atomicAdd(g_sigma, t);
    }
}

}  // namespace

IA_EXPORT int ia_gaussian_histogram(int64_t n, const float* x, const float* sigma, int bins, float vmin, float vmax, float* out,
                                    ia_stream_t stream)
{
    IA_REQUIRE(bins >= 1 && bins <= GH_MAX_BINS, "ia_gaussian_histogram: 1 <= bins <= 32");
    if (n == 0) return IA_OK;
    int grid = ia::cdiv(n, THREADS);
    if (grid > 1024) grid = 1024;
    gauss_hist_kernel<<<grid, THREADS, 0, (hipStream_t)stream>>>(n, x, sigma, bins, vmin, (vmax - vmin) / (float)bins, out);
    return ia::check_launch("ia_gaussian_histogram");
}

IA_EXPORT int ia_gaussian_histogram_bwd(int64_t n, const float* x, const float* sigma, int bins, float vmin, float vmax,
                                        const float* g_out, float* g_x, float* g_sigma, ia_stream_t stream)
{
    IA_REQUIRE(bins >= 1 && bins <= GH_MAX_BINS, "ia_gaussian_histogram_bwd: 1 <= bins <= 32");
    if (n == 0) return IA_OK;
    int grid = ia::cdiv(n, THREADS);
    if (grid > 1024) grid = 1024;
    gauss_hist_bwd_kernel<<<grid, THREADS, 0, (hipStream_t)stream>>>(n, x, sigma, bins, vmin, (vmax - vmin) / (float)bins, g_out, g_x,
                                                                     g_sigma);
    return ia::check_launch("ia_gaussian_histogram_bwd");
}

IA_EXPORT int ia_morton_keys(int64_t n, const float* pts, const float* origin_host3, float inv_cell, int32_t* keys, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    morton_keys_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, pts, origin_host3[0], origin_host3[1],
                                                                                 origin_host3[2], inv_cell, keys);
    return ia::check_launch("ia_morton_keys");
}

IA_EXPORT int ia_gather_rows3(int64_t n, const float* src, const int64_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    gather_rows3_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_gather_rows3");
}

IA_EXPORT int ia_scatter_f32(int64_t n, const float* src, const int64_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    scatter_f32_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_scatter_f32");
}

IA_EXPORT int ia_vi_layout(int64_t n_rays, int spp, const int32_t* rpi, const int32_t* bg_cnt, int32_t* fg_ray_cnt,
                           ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    vi_layout_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, (hipStream_t)stream>>>(n_rays, spp, rpi, bg_cnt, fg_ray_cnt);
    return ia::check_launch("ia_vi_layout");
}

IA_EXPORT int ia_vi_gather(int64_t n_rays, const int32_t* rpi, const int32_t* fg_ray_cnt, const int32_t* fg_start,
                           const float* ts, const int64_t* sampled_idx, const int32_t* fg_cnt, const float* weights,
                           const float* rays_o, const float* rays_d, const float* normals, const float* albedo,
                           const float* roughness, const float* metallic, int32_t* fg_src, int32_t* fg_ray, float* positions,
                           float* view_dirs, float* o_normals, float* o_albedo, float* o_roughness, float* o_metallic,
                           float* o_weights, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(n_rays < ((int64_t)1 << 31), "too many rays");
    vi_gather_kernel<<<(int)n_rays, THREADS, 0, (hipStream_t)stream>>>(n_rays, rpi, fg_ray_cnt, fg_start, ts, sampled_idx, fg_cnt,
                                                                      weights, rays_o, rays_d, normals, albedo, roughness,
                                                                      metallic, fg_src, fg_ray, positions, view_dirs, o_normals,
                                                                      o_albedo, o_roughness, o_metallic, o_weights);
    return ia::check_launch("ia_vi_gather");
}

IA_EXPORT int ia_vi_gather_bwd(int64_t S, const int32_t* fg_cnt, const int32_t* fg_off, const float* g_normals_fg,
                               const float* g_albedo_fg, const float* g_roughness_fg, const float* g_metallic_fg,
                               const float* g_weights_fg, float* g_normals, float* g_albedo, float* g_roughness,
                               float* g_metallic, float* g_weights, ia_stream_t stream)
{
    if (S == 0) return IA_OK;
    const int64_t waves = (S + 63) / 64;
    vi_gather_bwd_kernel<<<ia::cdiv(waves * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        S, fg_cnt, fg_off, g_normals_fg, g_albedo_fg, g_roughness_fg, g_metallic_fg, g_weights_fg, g_normals, g_albedo,
        g_roughness, g_metallic, g_weights);
    return ia::check_launch("ia_vi_gather_bwd");
}

IA_EXPORT int ia_vi_composite(int64_t n_rays, const int32_t* rpi, const int32_t* fg_ray_cnt, const int32_t* fg_start,
                              const int32_t* bg_cnt, const float* weights_fg, const float* Lo, const float* transmittance,
                              const float* background, const float* background_rays, float* rgb, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(n_rays < ((int64_t)1 << 25), "ia_vi_composite: n_rays must stay below 2^25 (one wave per ray, 32-bit grid)");
    vi_composite_kernel<<<ia::cdiv(n_rays * 64, THREADS), THREADS, 0, (hipStream_t)stream>>>(
        n_rays, rpi, fg_ray_cnt, fg_start, bg_cnt, weights_fg, Lo, transmittance, background, background_rays, rgb);
    return ia::check_launch("ia_vi_composite");
}

IA_EXPORT int ia_vi_composite_bwd(int64_t n_rays, int64_t F, const int32_t* rpi, const int32_t* bg_cnt, const int32_t* fg_ray,
                                  const float* weights_fg, const float* Lo, const float* background, const float* background_rays,
                                  const float* g_rgb, float* g_weights_fg, float* g_Lo, float* g_transmittance, ia_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (F > 0) vi_composite_bwd_kernel<<<ia::cdiv(F, THREADS), THREADS, 0, s>>>(F, fg_ray, weights_fg, Lo, g_rgb, g_weights_fg, g_Lo);
    if (n_rays > 0 && g_transmittance)
        vi_composite_bwd_T_kernel<<<ia::cdiv(n_rays, THREADS), THREADS, 0, s>>>(n_rays, rpi, bg_cnt, background, background_rays, g_rgb,
                                                                                g_transmittance);
    return ia::check_launch("ia_vi_composite_bwd");
}

IA_EXPORT int ia_light_shuffle(int64_t n_rays, int spp, const int32_t* fg_ray_cnt, const int32_t* fg_start, const float* u,
                               int32_t* shuffled, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(spp >= 1 && spp <= 4096, "ia_light_shuffle: 1 <= samples_per_pixel <= 4096");
    IA_REQUIRE(n_rays < ((int64_t)1 << 31), "ia_light_shuffle: n_rays must stay below 2^31 (one workgroup per ray)");
    int np = 1;
    while (np < spp) np <<= 1;
    if (np < 2) np = 2;
    light_shuffle_kernel<<<(int)n_rays, THREADS, (size_t)np * 8, (hipStream_t)stream>>>(n_rays, spp, np, fg_ray_cnt, fg_start, u,
                                                                                       shuffled);
    return ia::check_launch("ia_light_shuffle");
}

IA_EXPORT int ia_envlight_sample(int64_t k, const float* u, const double* cdf, int env_h, int env_w, const float* rot,
                                 float* dirs, ia_stream_t stream)
{
    if (k == 0) return IA_OK;
    IA_REQUIRE(env_h > 0 && env_w > 0, "environment map must be non-empty");
    const int64_t n = (int64_t)env_h * env_w;
    int log2b = 6;                                       // blocks of >= 64 entries, at most 4096 of them (32 KB of LDS)
    while (((n + ((int64_t)1 << log2b) - 1) >> log2b) > 4096) log2b++;
    const int nb = (int)((n + ((int64_t)1 << log2b) - 1) >> log2b);
    envlight_sample_kernel<<<ia::cdiv(k, ES_PER_WG), THREADS, (size_t)nb * sizeof(double), (hipStream_t)stream>>>(k, u, cdf, env_h, env_w, log2b,
                                                                                                                  rot, dirs);
    return ia::check_launch("ia_envlight_sample");
}

IA_EXPORT int ia_secondary_mask(int64_t F, const float* normals, const float* dirs, const int32_t* dir_index, int32_t* flag,
                                ia_stream_t stream)
{
    if (F == 0) return IA_OK;
    cos_mask_kernel<<<ia::cdiv(F, THREADS), THREADS, 0, (hipStream_t)stream>>>(F, normals, dirs, dir_index, flag);
    return ia::check_launch("ia_secondary_mask");
}

IA_EXPORT int ia_secondary_compact(int64_t F, const int32_t* flag, const int32_t* slot, const float* positions, const float* dirs,
                                   const int32_t* dir_index, float* rays_o, float* rays_d, int32_t* src, float* dense_dirs,
                                   ia_stream_t stream)
{
    if (F == 0) return IA_OK;
    compact_rays_kernel<<<ia::cdiv(F, THREADS), THREADS, 0, (hipStream_t)stream>>>(F, flag, slot, positions, dirs, dir_index, rays_o,
                                                                                  rays_d, src, dense_dirs);
    return ia::check_launch("ia_secondary_compact");
}

IA_EXPORT int ia_secondary_scatter(int64_t M, const int32_t* src, const float* transmittance, const float* rgb,
                                   float* dense_transmittance, float* dense_rgb, ia_stream_t stream)
{
    if (M == 0) return IA_OK;
    scatter_secondary_kernel<<<ia::cdiv(M, THREADS), THREADS, 0, (hipStream_t)stream>>>(M, src, transmittance, rgb,
                                                                                       dense_transmittance, dense_rgb);
    return ia::check_launch("ia_secondary_scatter");
}

IA_EXPORT int ia_secondary_gather_dense(int64_t F, const int32_t* flag, const int32_t* slot, const float* transmittance, const float* rgb,
                                        float* dense_transmittance, float* dense_rgb, ia_stream_t stream)
{
    if (F == 0) return IA_OK;
    IA_REQUIRE(flag && slot && dense_transmittance && dense_rgb, "null pointer");
    gather_secondary_kernel<<<ia::cdiv(F, THREADS), THREADS, 0, (hipStream_t)stream>>>(F, flag, slot, transmittance, rgb, dense_transmittance,
                                                                                      dense_rgb);
    return ia::check_launch("ia_secondary_gather_dense");
}

IA_EXPORT int ia_vi_indices(int64_t n_rays, int spp, const int32_t* rpi, const int32_t* fg_ray_cnt, const int32_t* fg_start,
                            const int32_t* bg_cnt, const int64_t* sampled_idx, const int32_t* fg_cnt, const float* weights,
                            const float* transmittance, int64_t* fg_indices, int64_t* bg_indices, int64_t* ray_indices,
                            float* resampled_weights, ia_stream_t stream)
{
    if (n_rays == 0) return IA_OK;
    IA_REQUIRE(n_rays < ((int64_t)1 << 31), "ia_vi_indices: n_rays must stay below 2^31 (one workgroup per ray)");
    vi_indices_kernel<<<(int)n_rays, THREADS, 0, (hipStream_t)stream>>>(n_rays, spp, rpi, fg_ray_cnt, fg_start, bg_cnt, sampled_idx,
                                                                       fg_cnt, weights, transmittance, fg_indices, bg_indices,
                                                                       ray_indices, resampled_weights);
    return ia::check_launch("ia_vi_indices");
}
