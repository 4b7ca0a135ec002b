// stepops.hip -- the small per-step operators of a training step, one launch each way.
//
// A 4096-ray training batch (the reference's own: configs/sampler/edge.yaml:2, configs/config.yaml:46-48) is bound by the HOST's launch
// rate, not by any kernel: profiles/r06_launch_audit_before.json counts 790 device launches per step of which 595 are ATen element-wise /
// fill / reduce kernels issued by chains of torch operators (and, for the differentiable ones, by their autograd backward chains).  The
// operators below are those chains as kernels (host orchestration):
//
//   ia_normalize_points           (x - center) / scale + 0.5                                models/rf/geometry.py:155, radiance.py:115
//   ia_effective_weights(_bwd)    weight norm / Lipschitz normalisation + level masks +      models/network_utils.py:201-244 (weight_norm),
//                                 the kernels' column order, per linear layer                 :396-403 (LipshitzMLP), :79-100 (level mask)
//   ia_sg_image(_bwd)             spherical-Gaussian lobes -> equirectangular image           lib/torch_pbr EnvironmentLightSG.generate_image
//                                                                                            (call site models/intrinsic_avatar.py:281-305)
//   ia_envlight_pdf_tables        luminance x sin(theta) -> pmf (fp32) and its running sum    emitter.update_pdf (:777-781)
//   ia_uniform_sphere_stratified  one jittered direction per equal-area stratum               emitter.sample_uniform_sphere_stratified (:680-689)
//   ia_material_affine(_bwd)      sigmoid outputs -> albedo / roughness / metallic ranges     models/pbr/material.py:44-50
//   ia_phys_loss(_bwd)            L1 rgb + L1 rgb_phys + BCE mask + eikonal mean -> loss      systems/intrinsic_avatar.py:165-252
//   ia_edge_min_sdf               min(sdf_left, sdf_right) on the left edges, 1e10 elsewhere  coarse_alpha_fn, models/intrinsic_avatar.py:980-990
//
// Arithmetic follows the torch expressions they replace operation by operation (no fma contraction in this translation unit); reductions
// are ordered (fixed tree per workgroup, partials summed in index order): results do not depend on the launch.
#include "ia_common.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide ordered sum of one float per thread (blockDim.x a multiple of 64, <= 1024); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh /* [16] */)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; w++) t += sh[w];
    return t;
}
__device__ __forceinline__ double block_sum_d(double v, double* sh /* [16] */)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum_d(v);
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; w++) t += sh[w];
    return t;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // torch.nn.functional.softplus (beta 1, threshold 20)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ normalize_points
__global__ __launch_bounds__(256) void normalize_points_kernel(int64_t n3, const float* __restrict__ x, const float* __restrict__ center,
                                                                const float* __restrict__ scale, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const int a = (int)(i % 3);
    out[i] = (x[i] - center[a]) / scale[a] + 0.5f;
}

// ------------------------------------------------------------------------------------------------ effective weights
// One wave per output row.  src[j] = column of the parameter that lands in output column j, mul[j] = its mask (NULL: 1).
//   mode 0  plain          out[m,j] = v[m,src j] * mul j
//   mode 1  weight norm    out[m,j] = g[m] * v[m,src j] / |v[m,:]| * mul j                      (network_utils.py:201-244)
//   mode 2  Lipschitz      out[m,j] = v[m,src j] * min(softplus(c) / sum_k |v[m,k]|, 1) * mul j  (network_utils.py:396-403; c: one scalar)
__global__ __launch_bounds__(256) void effective_weights_kernel(int mode, int M, int N, const float* __restrict__ g, const float* __restrict__ v,
                                                                 const int* __restrict__ src, const float* __restrict__ mul, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* row = v + (int64_t)m * N;
    float acc = 0.f;
    for (int k = lane; k < N; k += 64) acc += (mode == 1) ? row[k] * row[k] : fabsf(row[k]);
    acc = wave_sum(acc);
    float f = 1.f;
    if (mode == 1) f = g[m] / sqrtf(acc);
    else if (mode == 2) f = fminf(softplus_f(g[0]) / acc, 1.f);
    for (int j = lane; j < N; j += 64) {
        const int k = src ? src[j] : j;
        float w = row[k];
        if (mode == 1) w = g[m] * w / sqrtf(acc);          // torch: g * v / v.norm(dim=1, keepdim=True), left to right
        else if (mode == 2) w = w * f;
        if (mul) w = w * mul[j];
        out[(int64_t)m * N + j] = w;
    }
}

// backward of the above: g_out [M,N] -> g_v [M,N] (parameter column order), g_g [M] (mode 1) or g_g [1] (mode 2: summed over the rows in
// row order by ONE workgroup -- M <= 64).  Launched with one workgroup of 1024 threads (16 waves taking rows in turns).
__global__ __launch_bounds__(1024) void effective_weights_bwd_kernel(int mode, int M, int N, const float* __restrict__ g, const float* __restrict__ v,
                                                                      const int* __restrict__ src, const float* __restrict__ mul,
                                                                      const float* __restrict__ g_out, float* __restrict__ g_v, float* __restrict__ g_g)
{
    __shared__ float row_gc[64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int m = wid; m < M; m += nw) {
        const float* row = v + (int64_t)m * N;
        const float* go = g_out + (int64_t)m * N;
        float acc = 0.f, dot = 0.f;          // acc: |v|^2 or sum |v|; dot: sum_k u_k v_k with u = g_out pulled back to the parameter's columns
        for (int k = lane; k < N; k += 64) acc += (mode == 1) ? row[k] * row[k] : fabsf(row[k]);
        for (int j = lane; j < N; j += 64) {
            const int k = src ? src[j] : j;
            const float u = mul ? go[j] * mul[j] : go[j];
            dot += u * row[k];
        }
        acc = wave_sum(acc);
        dot = wave_sum(dot);
        if (mode == 0) {
            for (int j = lane; j < N; j += 64) g_v[(int64_t)m * N + (src ? src[j] : j)] = mul ? go[j] * mul[j] : go[j];
        } else if (mode == 1) {
            const float nrm = sqrtf(acc), gm = g[m];
            for (int j = lane; j < N; j += 64) {
                const int k = src ? src[j] : j;
                const float u = mul ? go[j] * mul[j] : go[j];
                g_v[(int64_t)m * N + k] = gm / nrm * (u - row[k] * (dot / acc));
            }
            if (lane == 0) g_g[m] = dot / nrm;
        } else {
            const float s = softplus_f(g[0]), r = s / acc;
            const bool active = r <= 1.f;          // torch.clamp(max=1) passes the gradient where input <= max
            const float f = fminf(r, 1.f);
            for (int j = lane; j < N; j += 64) {
                const int k = src ? src[j] : j;
                const float u = mul ? go[j] * mul[j] : go[j];
                const float w = row[k];
                const float sg = (w > 0.f) ? 1.f : ((w < 0.f) ? -1.f : 0.f);
                g_v[(int64_t)m * N + k] = u * f + (active ? -(dot * s / (acc * acc)) * sg : 0.f);
            }
            if (lane == 0) row_gc[m] = active ? dot / acc * sigmoid_f(g[0]) : 0.f;
        }
    }
    if (mode == 2) {
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int m = 0; m < M; m++) t += row_gc[m];
            g_g[0] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ SG environment image
// direction of pixel (row, col) of an H x W equirectangular image -- the convention of ia_envlight_eval / EnvironmentLightSG._dirs
__device__ __forceinline__ void equirect_dir(int row, int col, int H, int W, float d[3])
{
    const float v = ((float)row + 0.5f) / (float)H, u = ((float)col + 0.5f) / (float)W;
    const float th = v * PI_F, ph = (u - 0.5f) * 2.f * PI_F;
    d[0] = sinf(th) * sinf(ph);
    d[1] = cosf(th);
    d[2] = -sinf(th) * cosf(ph);
}

struct SgLobe { float xi[3], lam, s[3], inv_norm; };

__device__ __forceinline__ void load_lobes(int K, const float* axis, const float* log_lambda, const float* mu, SgLobe* sh)
{
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float a0 = axis[3 * k], a1 = axis[3 * k + 1], a2 = axis[3 * k + 2];
        const float nrm = fmaxf(sqrtf(a0 * a0 + a1 * a1 + a2 * a2), 1e-12f);      // torch.nn.functional.normalize: x / max(|x|, eps)
        sh[k].xi[0] = a0 / nrm; sh[k].xi[1] = a1 / nrm; sh[k].xi[2] = a2 / nrm;
        sh[k].inv_norm = 1.f / nrm;
        sh[k].lam = expf(log_lambda[k]);
        for (int c = 0; c < 3; c++) sh[k].s[c] = softplus_f(mu[3 * k + c]);
    }
    __syncthreads();
}

constexpr int SG_MAX_LOBES = 256;

__global__ __launch_bounds__(256) void sg_image_kernel(int K, int H, int W, const float* __restrict__ axis, const float* __restrict__ log_lambda,
                                                        const float* __restrict__ mu, float* __restrict__ out)
{
    __shared__ SgLobe lobes[SG_MAX_LOBES];
    load_lobes(K, axis, log_lambda, mu, lobes);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    float d[3];
    equirect_dir(p / W, p % W, H, W, d);
    float r = 0.f, g = 0.f, b = 0.f;
    for (int k = 0; k < K; k++) {
        const float c = d[0] * lobes[k].xi[0] + d[1] * lobes[k].xi[1] + d[2] * lobes[k].xi[2];
        const float w = expf(lobes[k].lam * (c - 1.f));
        r += w * lobes[k].s[0]; g += w * lobes[k].s[1]; b += w * lobes[k].s[2];
    }
    out[3 * p] = r; out[3 * p + 1] = g; out[3 * p + 2] = b;
}

// backward, stage 1: workgroup (k, chunk) sums lobe k's seven pixel sums over its share of the pixels -> partial [chunks][K][7]
//   0..2  sum_p w g_p            (d / d softplus(mu_k))
//   3     sum_p (g_p . s_k) w (c - 1)          (d / d lambda_k)
//   4..6  sum_p (g_p . s_k) w lambda_k d_p     (d / d xi_k)
__global__ __launch_bounds__(256) void sg_image_bwd_partial_kernel(int K, int H, int W, int chunks, const float* __restrict__ axis,
                                                                    const float* __restrict__ log_lambda, const float* __restrict__ mu,
                                                                    const float* __restrict__ g_img, float* __restrict__ partial)
{
    __shared__ float sh[16];
    const int k = blockIdx.x, chunk = blockIdx.y;
    const float a0 = axis[3 * k], a1 = axis[3 * k + 1], a2 = axis[3 * k + 2];
    const float nrm = fmaxf(sqrtf(a0 * a0 + a1 * a1 + a2 * a2), 1e-12f);
    const float xi[3] = {a0 / nrm, a1 / nrm, a2 / nrm};
    const float lam = expf(log_lambda[k]);
    const float s[3] = {softplus_f(mu[3 * k]), softplus_f(mu[3 * k + 1]), softplus_f(mu[3 * k + 2])};
    const int P = H * W, per = (P + chunks - 1) / chunks, p0 = chunk * per, p1 = min(P, p0 + per);
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        float d[3];
        equirect_dir(p / W, p % W, H, W, d);
        const float c = d[0] * xi[0] + d[1] * xi[1] + d[2] * xi[2];
        const float w = expf(lam * (c - 1.f));
        const float g0 = g_img[3 * p], g1 = g_img[3 * p + 1], g2 = g_img[3 * p + 2];
        const float gw = (g0 * s[0] + g1 * s[1] + g2 * s[2]) * w;
        acc[0] += w * g0; acc[1] += w * g1; acc[2] += w * g2;
        acc[3] += gw * (c - 1.f);
        const float gc = gw * lam;
        acc[4] += gc * d[0]; acc[5] += gc * d[1]; acc[6] += gc * d[2];
    }
    for (int j = 0; j < 7; j++) {
        const float t = block_sum(acc[j], sh);
        if (threadIdx.x == 0) partial[((int64_t)chunk * K + k) * 7 + j] = t;
    }
}

// stage 2: one thread per lobe sums the chunks in order and applies the parameterisations
__global__ void sg_image_bwd_final_kernel(int K, int chunks, const float* __restrict__ axis, const float* __restrict__ log_lambda,
                                          const float* __restrict__ mu, const float* __restrict__ partial, float* __restrict__ g_axis,
                                          float* __restrict__ g_log_lambda, float* __restrict__ g_mu)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < chunks; c++)
        for (int j = 0; j < 7; j++) t[j] += partial[((int64_t)c * K + k) * 7 + j];
    for (int c = 0; c < 3; c++) g_mu[3 * k + c] = t[c] * sigmoid_f(mu[3 * k + c]);
    g_log_lambda[k] = t[3] * expf(log_lambda[k]);
    const float a[3] = {axis[3 * k], axis[3 * k + 1], axis[3 * k + 2]};
    const float n2 = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n2 > 1e-12f) {           // xi = a / |a|:  d xi = (I - xi xi^T) / |a|
        const float xi[3] = {a[0] / n2, a[1] / n2, a[2] / n2};
        const float dt = xi[0] * t[4] + xi[1] * t[5] + xi[2] * t[6];
        for (int c = 0; c < 3; c++) g_axis[3 * k + c] = (t[4 + c] - xi[c] * dt) / n2;
    } else {                     // clamped denominator: xi = a / eps
        for (int c = 0; c < 3; c++) g_axis[3 * k + c] = t[4 + c] / 1e-12f;
    }
}

// ------------------------------------------------------------------------------------------------ envlight pdf tables
// EnvironmentLightTensor.update_pdf: w = max(luminance, 0) * sin(theta) in double; pmf = float(w / sum w); cdf = running sum of the
// fp32 pmf in double.  Three element-parallel launches over tiles of 1024 texels (a single workgroup streaming the 256 x 512 training
// light was latency-bound on one CU: 180 us of a 14 ms step); every sum is ordered: per-tile tree, tiles in index order.
constexpr int PDF_TILE = 1024;

__global__ __launch_bounds__(256) void envlight_w_kernel(int P, int W, int H, const float* __restrict__ base, double* __restrict__ w_out,
                                                         double* __restrict__ tile_sum)
{
    __shared__ double sh[16];
    const int p0 = blockIdx.x * PDF_TILE + threadIdx.x * 4;
    double local = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = p0 + k;
        if (p < P) {
            const float lum = fmaxf(0.2126f * base[3 * p] + 0.7152f * base[3 * p + 1] + 0.0722f * base[3 * p + 2], 0.f);
            const float sin_t = sinf(((float)(p / W) + 0.5f) * PI_F / (float)H);
            const double w = (double)lum * (double)sin_t;
            w_out[p] = w;
            local += w;
        }
    }
    const double t = block_sum_d(local, sh);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = t;
}

// pmf of a tile + the tile's inclusive running sum of the fp32 pmf (double) + the tile's pmf sum
__global__ __launch_bounds__(256) void envlight_pmf_kernel(int P, int n_tiles, const double* __restrict__ tile_sum, double* __restrict__ w_cdf,
                                                           float* __restrict__ pmf, double* __restrict__ tile_pmf_sum)
{
    __shared__ double sh[16];
    double total = 0.0;
    for (int t = 0; t < n_tiles; t++) total += tile_sum[t];          // the same ordered sum in every workgroup
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p0 = blockIdx.x * PDF_TILE + tid * 4;
    double q[4], run = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float f = 0.f;
        if (p0 + k < P) {
            f = (float)(w_cdf[p0 + k] / total);
            pmf[p0 + k] = f;
        }
        run += (double)f;
        q[k] = run;
    }
    double inc = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sh[wid] = inc;
    __syncthreads();
    double off_w = 0.0, tot = 0.0;
    for (int w = 0; w < 4; w++) {
        if (w < wid) off_w += sh[w];
        tot += sh[w];
    }
    const double basev = off_w + (inc - run);
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (p0 + k < P) w_cdf[p0 + k] = basev + q[k];
    if (tid == 0) tile_pmf_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void envlight_cdf_offset_kernel(int P, const double* __restrict__ tile_pmf_sum, double* __restrict__ cdf)
{
    double off = 0.0;
    for (int t = 0; t < (int)blockIdx.x; t++) off += tile_pmf_sum[t];          // tiles in index order
    const int p0 = blockIdx.x * PDF_TILE + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (p0 + k < P) cdf[p0 + k] = off + cdf[p0 + k];
}

// ------------------------------------------------------------------------------------------------ stratified sphere
__global__ void uniform_sphere_stratified_kernel(int n_theta, int n_phi, const float* __restrict__ u, float* __restrict__ dirs,
                                                 float* __restrict__ inv_pdf)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_theta * n_phi) return;
    const float i = (float)(k / n_phi), j = (float)(k % n_phi);
    const float z = 1.0f - 2.0f * (i + u[2 * k]) / (float)n_theta;
    const float phi = 2.0f * PI_F * (j + u[2 * k + 1]) / (float)n_phi;
    const float r = sqrtf(fmaxf(1.0f - z * z, 0.f));
    dirs[3 * k] = r * cosf(phi);
    dirs[3 * k + 1] = r * sinf(phi);
    dirs[3 * k + 2] = z;
    inv_pdf[k] = 4.0f * PI_F;
}

// ------------------------------------------------------------------------------------------------ material affine
__global__ __launch_bounds__(256) void material_affine_kernel(int64_t n, const float* __restrict__ m, float as, float ab, float rs, float rb,
                                                               float ms, float mb, float* __restrict__ albedo, float* __restrict__ rough,
                                                               float* __restrict__ metal)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = m + 5 * i;
    albedo[3 * i] = r[0] * as + ab; albedo[3 * i + 1] = r[1] * as + ab; albedo[3 * i + 2] = r[2] * as + ab;
    rough[i] = r[3] * rs + rb;
    metal[i] = r[4] * ms + mb;
}
__global__ __launch_bounds__(256) void material_affine_bwd_kernel(int64_t n, const float* __restrict__ g_alb, const float* __restrict__ g_rgh,
                                                                   const float* __restrict__ g_mtl, float as, float rs, float ms,
                                                                   float* __restrict__ g_m)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* o = g_m + 5 * i;
    for (int c = 0; c < 3; c++) o[c] = g_alb ? g_alb[3 * i + c] * as : 0.f;
    o[3] = g_rgh ? g_rgh[i] * rs : 0.f;
    o[4] = g_mtl ? g_mtl[i] * ms : 0.f;
}

// ------------------------------------------------------------------------------------------------ loss
// terms[0] mean |comp_rgb - target|, [1] mean |comp_rgb_phys - target|, [2] BCE(clamp(opacity, 1e-3, 1 - 1e-3), mask), [3] eikonal sum,
// [4] loss = t0 + lambda_phys t1 + lambda_mask t2 + lambda_eik t3 / eik_denom.  One workgroup, ordered sums.
__device__ __forceinline__ void phys_loss_terms_of_ray(int64_t i, const float* __restrict__ rgb, const float* __restrict__ rgb_phys,
                                                       const float* __restrict__ opacity, const float* __restrict__ target,
                                                       const float* __restrict__ mask, float& a, float& b, float& c)
{
    for (int k = 0; k < 3; k++) {
        const float t = target[3 * i + k];
        a += fabsf(rgb[3 * i + k] - t);
        if (rgb_phys) b += fabsf(rgb_phys[3 * i + k] - t);
    }
    if (mask) {
        const float x = fminf(fmaxf(opacity[i], 1e-3f), 1.f - 1e-3f), t = mask[i];
        c += -(t * fmaxf(logf(x), -100.f) + (1.f - t) * fmaxf(logf(1.f - x), -100.f));      // torch's BCE clamps its logs at -100
    }
}

// large frames: per-workgroup partial sums [n_wg][3] (1024 rays each), summed in index order by phys_loss_kernel
__global__ __launch_bounds__(256) void phys_loss_partial_kernel(int64_t n, const float* __restrict__ rgb, const float* __restrict__ rgb_phys,
                                                                const float* __restrict__ opacity, const float* __restrict__ target,
                                                                const float* __restrict__ mask, float* __restrict__ partial)
{
    __shared__ float sh[16];
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = 0; k < 4; k++) {
        const int64_t i = (int64_t)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (i < n) phys_loss_terms_of_ray(i, rgb, rgb_phys, opacity, target, mask, a, b, c);
    }
    const float sa = block_sum(a, sh), sb = block_sum(b, sh), sc = block_sum(c, sh);
    if (threadIdx.x == 0) { partial[3 * blockIdx.x] = sa; partial[3 * blockIdx.x + 1] = sb; partial[3 * blockIdx.x + 2] = sc; }
}

__global__ __launch_bounds__(1024) void phys_loss_kernel(int64_t n, const float* __restrict__ rgb, const float* __restrict__ rgb_phys,
                                                          const float* __restrict__ opacity, const float* __restrict__ target,
                                                          const float* __restrict__ mask, const float* __restrict__ eik_part, int eik_k,
                                                          const float* __restrict__ partial, int n_partial,
                                                          float lambda_phys, float lambda_mask, float lambda_eik, float eik_denom,
                                                          float* __restrict__ terms)
{
    __shared__ float sh[16];
    float a = 0.f, b = 0.f, c = 0.f;
    if (partial) {
        for (int k = threadIdx.x; k < n_partial; k += blockDim.x) { a += partial[3 * k]; b += partial[3 * k + 1]; c += partial[3 * k + 2]; }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) phys_loss_terms_of_ray(i, rgb, rgb_phys, opacity, target, mask, a, b, c);
    }
    float e = 0.f;
    for (int k = threadIdx.x; k < eik_k; k += blockDim.x) e += eik_part[2 * k];          // (eik_k = 0 without the term)
    const float sa = block_sum(a, sh), sb = block_sum(b, sh), sc = block_sum(c, sh);
    e = block_sum(e, sh);
    if (threadIdx.x == 0) {
        const float t0 = sa / (float)(3 * n), t1 = sb / (float)(3 * n), t2 = sc / (float)n;
        terms[0] = t0; terms[1] = t1; terms[2] = t2; terms[3] = e;
        float loss = t0;
        if (eik_part) loss = loss + lambda_eik * e / eik_denom;
        if (mask) loss = loss + lambda_mask * t2;
        if (rgb_phys) loss = loss + lambda_phys * t1;
        terms[4] = loss;
    }
}

__global__ __launch_bounds__(256) void phys_loss_bwd_kernel(int64_t n, const float* __restrict__ rgb, const float* __restrict__ rgb_phys,
                                                             const float* __restrict__ opacity, const float* __restrict__ target,
                                                             const float* __restrict__ mask, const float* __restrict__ g_loss, float lambda_phys,
                                                             float lambda_mask, float lambda_eik, float eik_denom, float* __restrict__ g_rgb,
                                                             float* __restrict__ g_rgb_phys, float* __restrict__ g_opacity, float* __restrict__ g_eik)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float g = g_loss[0];
    if (i == 0 && g_eik) g_eik[0] = g * lambda_eik / eik_denom;
    if (i >= n) return;
    const float inv3n = 1.f / (float)(3 * n);
    for (int k = 0; k < 3; k++) {
        const float t = target[3 * i + k];
        const float d = rgb[3 * i + k] - t;
        g_rgb[3 * i + k] = g * inv3n * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
        if (rgb_phys) {
            const float dp = rgb_phys[3 * i + k] - t;
            g_rgb_phys[3 * i + k] = g * lambda_phys * inv3n * ((dp > 0.f) ? 1.f : ((dp < 0.f) ? -1.f : 0.f));
        }
    }
    if (mask) {
        const float o = opacity[i];
        float go = 0.f;
        if (o >= 1e-3f && o <= 1.f - 1e-3f) {      // clamp passes the gradient inside its range
            const float t = mask[i];
            go = g * lambda_mask / (float)n * (o - t) / fmaxf((1.f - o) * o, 1e-12f);      // binary_cross_entropy_backward
        }
        g_opacity[i] = go;
    }
}

// ------------------------------------------------------------------------------------------------ world -> SMPL ray transform
// SNARFDeformer.transform_rays_w2s (snarf_deformer.py:128-147): o' = o R^T + t, d' = d R^T, near / far = |o'| -+ 1.  `variant` selects the
// summation form of the 3-term products (the reference does them as [n,3] x [3,3] GEMMs; which form equals the library's result bit for
// bit is established by tools/ray_transform_probe.py, and only that form is used): 0 fma chain k = 0,1,2; 1 separate products, left to
// right; 2 fma chain k = 2,1,0.
__device__ __forceinline__ float dot3_variant(const float* __restrict__ p, const float* __restrict__ r, int variant)
{
    if (variant == 0) return __fmaf_rn(p[2], r[2], __fmaf_rn(p[1], r[1], p[0] * r[0]));
    if (variant == 1) return __fadd_rn(__fadd_rn(__fmul_rn(p[0], r[0]), __fmul_rn(p[1], r[1])), __fmul_rn(p[2], r[2]));
    return __fmaf_rn(p[0], r[0], __fmaf_rn(p[1], r[1], p[2] * r[2]));
}

__global__ __launch_bounds__(256) void transform_rays_kernel(int64_t n, const float* __restrict__ rays, int stride, const float* __restrict__ w2s,
                                                             int variant, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + i * stride;
    float o[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o[a] = __fadd_rn(dot3_variant(r, w2s + 4 * a, variant & 3), w2s[4 * a + 3]);
        d[a] = dot3_variant(r + 3, w2s + 4 * a, variant & 3);
    }
    const int nv = variant >> 2;          // norm form: 0 separate products left to right, 1 fma chain k = 0,1,2, 2 fma chain k = 2,1,0
    float ss;
    if (nv == 0) ss = __fadd_rn(__fadd_rn(__fmul_rn(o[0], o[0]), __fmul_rn(o[1], o[1])), __fmul_rn(o[2], o[2]));
    else if (nv == 1) ss = __fmaf_rn(o[2], o[2], __fmaf_rn(o[1], o[1], o[0] * o[0]));
    else ss = __fmaf_rn(o[0], o[0], __fmaf_rn(o[1], o[1], o[2] * o[2]));
    const float dist = sqrtf(ss);
    float* q = out + i * 8;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; q[3] = d[0]; q[4] = d[1]; q[5] = d[2];
    q[6] = dist - 1.0f; q[7] = dist + 1.0f;
}

// ------------------------------------------------------------------------------------------------ edge min
__global__ __launch_bounds__(256) void edge_min_sdf_kernel(int64_t E, const float* __restrict__ sdf, const uint8_t* __restrict__ is_left,
                                                            float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const float nxt = sdf[i + 1 < E ? i + 1 : E - 1];
    out[i] = is_left[i] ? fminf(sdf[i], nxt) : 1e10f;
}

}  // namespace

// ================================================================================================ C ABI
IA_EXPORT int ia_normalize_points(int64_t n, const float* x, const float* center, const float* scale, float* out, ia_stream_t stream)
{
    if (n <= 0) return IA_OK;
    IA_REQUIRE(x && center && scale && out, "null pointer");
    normalize_points_kernel<<<ia::cdiv(3 * n, 256), 256, 0, (hipStream_t)stream>>>(3 * n, x, center, scale, out);
    return ia::check_launch("ia_normalize_points");
}

IA_EXPORT int ia_effective_weights(int mode, int M, int N, const float* g, const float* v, const int* src, const float* mul, float* out,
                                   ia_stream_t stream)
{
    IA_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (plain), 1 (weight norm) or 2 (Lipschitz)");
    IA_REQUIRE(M > 0 && N > 0 && v && out && (mode == 0 || g), "bad arguments");
    effective_weights_kernel<<<ia::cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(mode, M, N, g, v, src, mul, out);
    return ia::check_launch("ia_effective_weights");
}

IA_EXPORT int ia_effective_weights_bwd(int mode, int M, int N, const float* g, const float* v, const int* src, const float* mul,
                                       const float* g_out, float* g_v, float* g_g, ia_stream_t stream)
{
    IA_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (plain), 1 (weight norm) or 2 (Lipschitz)");
    IA_REQUIRE(M > 0 && N > 0 && v && g_out && g_v && (mode == 0 || (g && g_g)), "bad arguments");
    IA_REQUIRE(mode != 2 || M <= 64, "Lipschitz layers have at most 64 rows");
    effective_weights_bwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(mode, M, N, g, v, src, mul, g_out, g_v, g_g);
    return ia::check_launch("ia_effective_weights_bwd");
}

IA_EXPORT int ia_sg_image(int K, int H, int W, const float* axis, const float* log_lambda, const float* mu, float* out, ia_stream_t stream)
{
    IA_REQUIRE(K > 0 && K <= SG_MAX_LOBES, "1 .. 256 lobes");
    IA_REQUIRE(H > 0 && W > 0 && axis && log_lambda && mu && out, "bad arguments");
    sg_image_kernel<<<ia::cdiv((int64_t)H * W, 256), 256, 0, (hipStream_t)stream>>>(K, H, W, axis, log_lambda, mu, out);
    return ia::check_launch("ia_sg_image");
}

static constexpr int SG_BWD_CHUNKS = 32;
IA_EXPORT int64_t ia_sg_image_bwd_tmp_bytes(int K) { return (int64_t)SG_BWD_CHUNKS * K * 7 * 4; }

IA_EXPORT int ia_sg_image_bwd(int K, int H, int W, const float* axis, const float* log_lambda, const float* mu, const float* g_img, void* tmp,
                              float* g_axis, float* g_log_lambda, float* g_mu, ia_stream_t stream)
{
    IA_REQUIRE(K > 0 && K <= SG_MAX_LOBES, "1 .. 256 lobes");
    IA_REQUIRE(H > 0 && W > 0 && axis && log_lambda && mu && g_img && tmp && g_axis && g_log_lambda && g_mu, "bad arguments");
    sg_image_bwd_partial_kernel<<<dim3(K, SG_BWD_CHUNKS), 256, 0, (hipStream_t)stream>>>(K, H, W, SG_BWD_CHUNKS, axis, log_lambda, mu, g_img,
                                                                                            (float*)tmp);
    sg_image_bwd_final_kernel<<<ia::cdiv(K, 64), 64, 0, (hipStream_t)stream>>>(K, SG_BWD_CHUNKS, axis, log_lambda, mu, (const float*)tmp, g_axis,
                                                                                g_log_lambda, g_mu);
    return ia::check_launch("ia_sg_image_bwd");
}

IA_EXPORT int64_t ia_envlight_pdf_tables_tmp_bytes(int H, int W) { return 2 * (int64_t)(((int64_t)H * W + PDF_TILE - 1) / PDF_TILE) * 8 + 64; }

IA_EXPORT int ia_envlight_pdf_tables(int H, int W, const float* base, float* pmf, double* cdf, void* tmp, ia_stream_t stream)
{
    IA_REQUIRE(H > 0 && W > 0 && base && pmf && cdf && tmp, "bad arguments");
    const int64_t P64 = (int64_t)H * W;
    IA_REQUIRE(P64 < ((int64_t)1 << 30), "image too large");
    const int P = (int)P64, n_tiles = (P + PDF_TILE - 1) / PDF_TILE;
    IA_REQUIRE(n_tiles <= 4096, "at most 4096 tiles of 1024 texels (larger images: the caller's own reduction)");
    double* tile_sum = (double*)tmp;
    double* tile_pmf = tile_sum + n_tiles;
    hipStream_t s = (hipStream_t)stream;
    envlight_w_kernel<<<n_tiles, 256, 0, s>>>(P, W, H, base, cdf, tile_sum);
    envlight_pmf_kernel<<<n_tiles, 256, 0, s>>>(P, n_tiles, tile_sum, cdf, pmf, tile_pmf);
    envlight_cdf_offset_kernel<<<n_tiles, 256, 0, s>>>(P, tile_pmf, cdf);
    return ia::check_launch("ia_envlight_pdf_tables");
}

IA_EXPORT int ia_uniform_sphere_stratified(int n_theta, int n_phi, const float* u, float* dirs, float* inv_pdf, ia_stream_t stream)
{
    IA_REQUIRE(n_theta > 0 && n_phi > 0 && u && dirs && inv_pdf, "bad arguments");
    uniform_sphere_stratified_kernel<<<ia::cdiv((int64_t)n_theta * n_phi, 256), 256, 0, (hipStream_t)stream>>>(n_theta, n_phi, u, dirs, inv_pdf);
    return ia::check_launch("ia_uniform_sphere_stratified");
}

IA_EXPORT int ia_material_affine(int64_t n, const float* m, float albedo_scale, float albedo_bias, float roughness_scale, float roughness_bias,
                                 float metallic_scale, float metallic_bias, float* albedo, float* roughness, float* metallic, ia_stream_t stream)
{
    if (n <= 0) return IA_OK;
    IA_REQUIRE(m && albedo && roughness && metallic, "null pointer");
    material_affine_kernel<<<ia::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(n, m, albedo_scale, albedo_bias, roughness_scale, roughness_bias,
                                                                              metallic_scale, metallic_bias, albedo, roughness, metallic);
    return ia::check_launch("ia_material_affine");
}

IA_EXPORT int ia_material_affine_bwd(int64_t n, const float* g_albedo, const float* g_roughness, const float* g_metallic, float albedo_scale,
                                     float roughness_scale, float metallic_scale, float* g_m, ia_stream_t stream)
{
    if (n <= 0) return IA_OK;
    IA_REQUIRE(g_m, "null pointer");
    material_affine_bwd_kernel<<<ia::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(n, g_albedo, g_roughness, g_metallic, albedo_scale,
                                                                                  roughness_scale, metallic_scale, g_m);
    return ia::check_launch("ia_material_affine_bwd");
}

IA_EXPORT int64_t ia_phys_loss_tmp_bytes(int64_t n) { return n > 16384 ? ((n + 1023) / 1024) * 3 * 4 + 64 : 0; }

IA_EXPORT int ia_phys_loss(int64_t n, const float* comp_rgb, const float* comp_rgb_phys, const float* opacity, const float* target_rgb,
                           const float* target_mask, const float* eik_partials, int eik_k, float lambda_phys, float lambda_mask,
                           float lambda_eik, float eik_denom, float* terms, void* tmp, ia_stream_t stream)
{
    IA_REQUIRE(n > 0 && comp_rgb && target_rgb && terms, "bad arguments");
    IA_REQUIRE(!target_mask || opacity, "the mask term needs the opacity");
    hipStream_t s = (hipStream_t)stream;
    float* partial = nullptr;
    int n_partial = 0;
    if (n > 16384) {          // a full frame: workgroup partials first (one workgroup streaming 291 600 rays took 0.5 ms)
        IA_REQUIRE(tmp != nullptr, "tmp (ia_phys_loss_tmp_bytes) is required above 16384 rays");
        n_partial = (int)((n + 1023) / 1024);
        partial = (float*)tmp;
        phys_loss_partial_kernel<<<n_partial, 256, 0, s>>>(n, comp_rgb, comp_rgb_phys, opacity, target_rgb, target_mask, partial);
    }
    phys_loss_kernel<<<1, 1024, 0, s>>>(n, comp_rgb, comp_rgb_phys, opacity, target_rgb, target_mask, eik_partials, eik_k, partial, n_partial,
                                         lambda_phys, lambda_mask, lambda_eik, eik_denom, terms);
    return ia::check_launch("ia_phys_loss");
}

IA_EXPORT int ia_phys_loss_bwd(int64_t n, const float* comp_rgb, const float* comp_rgb_phys, const float* opacity, const float* target_rgb,
                               const float* target_mask, const float* g_loss, float lambda_phys, float lambda_mask, float lambda_eik,
                               float eik_denom, float* g_comp_rgb, float* g_comp_rgb_phys, float* g_opacity, float* g_eik_sum,
                               ia_stream_t stream)
{
    IA_REQUIRE(n > 0 && comp_rgb && target_rgb && g_loss && g_comp_rgb, "bad arguments");
    IA_REQUIRE((!comp_rgb_phys || g_comp_rgb_phys) && (!target_mask || (opacity && g_opacity)), "missing gradient buffer");
    phys_loss_bwd_kernel<<<ia::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(n, comp_rgb, comp_rgb_phys, opacity, target_rgb, target_mask, g_loss,
                                                                            lambda_phys, lambda_mask, lambda_eik, eik_denom, g_comp_rgb,
                                                                            g_comp_rgb_phys, g_opacity, g_eik_sum);
    return ia::check_launch("ia_phys_loss_bwd");
}

IA_EXPORT int ia_transform_rays_w2s(int64_t n, const float* rays, int ray_stride, const float* w2s, int variant, float* out, ia_stream_t stream)
{
    if (n <= 0) return IA_OK;
    IA_REQUIRE(rays && w2s && out && ray_stride >= 6 && variant >= 0 && variant <= 10, "bad arguments");
    transform_rays_kernel<<<ia::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(n, rays, ray_stride, w2s, variant, out);
    return ia::check_launch("ia_transform_rays_w2s");
}

IA_EXPORT int ia_edge_min_sdf(int64_t n_edges, const float* sdf, const uint8_t* is_left, float* out, ia_stream_t stream)
{
    if (n_edges <= 0) return IA_OK;
    IA_REQUIRE(sdf && is_left && out, "null pointer");
    edge_min_sdf_kernel<<<ia::cdiv(n_edges, 256), 256, 0, (hipStream_t)stream>>>(n_edges, sdf, is_left, out);
    return ia::check_launch("ia_edge_min_sdf");
}
