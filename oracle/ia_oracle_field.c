/*
 * ia_oracle_field.c -- CPU ORACLE (test infrastructure, NOT the product), part 2:
 * the neural-field queries of the render_step hot path.
 *
 *   - multiresolution hash grid  (tinycudann HashGrid; reference call sites
 *     models/network_utils.py:58-100, configs/geometry/progressive_hash_grid.yaml:9-24)
 *   - spherical harmonics deg 4  (tinycudann SphericalHarmonics;
 *     configs/radiance/progressive_hash_grid.yaml:17-19, models/rf/radiance.py:124-126)
 *   - the three small MLPs       (models/network_utils.py:201-244 VanillaMLP,
 *     :360-431 LipshitzMLP) evaluated with *effective* weights
 *   - Laplace density / alpha    (models/rf/density.py:25-30, models/intrinsic_avatar.py:390-394)
 *
 * tinycudann is NOT vendored in /root/reference: the hash-grid and SH
 * restatements follow the published Instant-NGP / tiny-cuda-nn definitions and are
 * "PARITY UNPINNED" against upstream; the MLP restatements are pinned against the
 * reference's own VanillaMLP / LipshitzMLP modules (tests/golden/mlp_*.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IA_API __attribute__((visibility("default")))

/* ---- hash grid geometry (tiny-cuda-nn grid.h: grid_scale / grid_resolution / offset table) ---- */
static float grid_scale(int level, float log2_per_level_scale, int base_resolution)
{
    return exp2f((float)level * log2_per_level_scale) * (float)base_resolution - 1.0f;
}
static uint32_t grid_resolution(float scale) { return (uint32_t)ceilf(scale) + 1u; }

/* offsets[n_levels+1] in *entries* (each entry = F floats). returns total entries. */
IA_API int64_t ia_ref_hashgrid_offsets(int n_levels, int log2_hashmap_size, int base_resolution,
                                       float per_level_scale, uint32_t *offsets, uint32_t *resolutions,
                                       float *scales)
{
    uint32_t offset = 0;
    float l2 = log2f(per_level_scale);
    for (int l = 0; l < n_levels; l++) {
        float sc = grid_scale(l, l2, base_resolution);
        uint32_t res = grid_resolution(sc);
        uint32_t max_params = 0xFFFFFFFFu / 2;
        uint32_t p = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
        p = (p + 7u) / 8u * 8u;
        uint32_t cap = 1u << log2_hashmap_size;
        if (p > cap) p = cap;
        offsets[l] = offset;
        if (resolutions) resolutions[l] = res;
        if (scales) scales[l] = sc;
        offset += p;
    }
    offsets[n_levels] = offset;
    return (int64_t)offset;
}

static inline uint32_t grid_index(uint32_t hashmap_size, uint32_t res, const uint32_t p[3])
{
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < 3 && stride <= hashmap_size; d++) {
        index += p[d] * stride;
        stride *= res;
    }
    if (hashmap_size < stride)
        index = (p[0] * 1u) ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u);
    return index % hashmap_size;
}

/* forward: x in [0,1]^3 -> out [n, L*F] (level-major), optional dy_dx [n, L*F, 3] */
IA_API int ia_ref_hashgrid_fwd(int64_t n, const float *x, const float *params, int n_levels, int F,
                               int log2_hashmap_size, int base_resolution, float per_level_scale,
                               float *out, float *dy_dx)
{
    uint32_t offsets[64]; uint32_t ress[64]; float scales[64];
    ia_ref_hashgrid_offsets(n_levels, log2_hashmap_size, base_resolution, per_level_scale, offsets, ress, scales);
    for (int64_t i = 0; i < n; i++) {
        for (int l = 0; l < n_levels; l++) {
            float sc = scales[l];
            uint32_t res = ress[l], hsize = offsets[l + 1] - offsets[l];
            const float *tab = params + (int64_t)offsets[l] * F;
            float pos[3]; uint32_t pg[3];
            for (int d = 0; d < 3; d++) {
                float p = fmaf(sc, x[i * 3 + d], 0.5f);
                float fl = floorf(p);
                pg[d] = (uint32_t)(int)fl;
                pos[d] = p - fl;
            }
            float acc[8] = {0}, dacc[8][3];
            memset(dacc, 0, sizeof(dacc));
            for (int c = 0; c < 8; c++) {
                float w = 1.0f; uint32_t pl[3];
                for (int d = 0; d < 3; d++) {
                    if ((c & (1 << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                uint32_t idx = grid_index(hsize, res, pl);
                for (int f = 0; f < F; f++) acc[f] += w * tab[(int64_t)idx * F + f];
                if (dy_dx) {
                    for (int gd = 0; gd < 3; gd++) {
                        float wd = ((c & (1 << gd)) == 0) ? -sc : sc;
                        for (int d = 0; d < 3; d++) {
                            if (d == gd) continue;
                            wd *= ((c & (1 << d)) == 0) ? (1.0f - pos[d]) : pos[d];
                        }
                        for (int f = 0; f < F; f++) dacc[f][gd] += wd * tab[(int64_t)idx * F + f];
                    }
                }
            }
            for (int f = 0; f < F; f++) {
                out[i * n_levels * F + l * F + f] = acc[f];
                if (dy_dx)
                    for (int gd = 0; gd < 3; gd++)
                        dy_dx[(i * n_levels * F + l * F + f) * 3 + gd] = dacc[f][gd];
            }
        }
    }
    return 0;
}

/* backward wrt params: grad_params[idx] += w * dL_dy  (grad_params zeroed by caller) */
IA_API int ia_ref_hashgrid_bwd_params(int64_t n, const float *x, const float *dL_dy, int n_levels, int F,
                                      int log2_hashmap_size, int base_resolution, float per_level_scale,
                                      float *grad_params)
{
    uint32_t offsets[64]; uint32_t ress[64]; float scales[64];
    ia_ref_hashgrid_offsets(n_levels, log2_hashmap_size, base_resolution, per_level_scale, offsets, ress, scales);
    for (int64_t i = 0; i < n; i++) {
        for (int l = 0; l < n_levels; l++) {
            float sc = scales[l];
            uint32_t res = ress[l], hsize = offsets[l + 1] - offsets[l];
            float *tab = grad_params + (int64_t)offsets[l] * F;
            float pos[3]; uint32_t pg[3];
            for (int d = 0; d < 3; d++) {
                float p = fmaf(sc, x[i * 3 + d], 0.5f);
                float fl = floorf(p);
                pg[d] = (uint32_t)(int)fl;
                pos[d] = p - fl;
            }
            for (int c = 0; c < 8; c++) {
                float w = 1.0f; uint32_t pl[3];
                for (int d = 0; d < 3; d++) {
                    if ((c & (1 << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                uint32_t idx = grid_index(hsize, res, pl);
                for (int f = 0; f < F; f++) tab[(int64_t)idx * F + f] += w * dL_dy[i * n_levels * F + l * F + f];
            }
        }
    }
    return 0;
}

/* spherical harmonics degree 4: d01 in [0,1]^3 (tcnn convention) -> 16 */
IA_API int ia_ref_sh4(int64_t n, const float *d01, float *out)
{
    for (int64_t i = 0; i < n; i++) {
        float x = d01[i * 3 + 0] * 2.f - 1.f, y = d01[i * 3 + 1] * 2.f - 1.f, z = d01[i * 3 + 2] * 2.f - 1.f;
        float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        float *o = out + i * 16;
        o[0] = 0.28209479177387814f;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
        o[4] = 1.0925484305920792f * xy;
        o[5] = -1.0925484305920792f * yz;
        o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[7] = -1.0925484305920792f * xz;
        o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[10] = 2.8906114426405538f * xy * z;
        o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[14] = 1.4453057213202769f * z * (x2 - y2);
        o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
    return 0;
}

/* activations */
static inline float softplus100(float x)
{
    /* torch.nn.Softplus(beta=100, threshold=20): network_utils.py:240-242 */
    float bx = 100.0f * x;
    return bx > 20.0f ? x : log1pf(expf(bx)) / 100.0f;
}
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* generic MLP forward with effective (already weight-normed / Lipschitz-normalised)
 * weights. Layers: dims[0] -> dims[1] -> ... -> dims[n_layers]; W_l row-major [out,in].
 * hidden_act: 0 ReLU, 1 Softplus(beta=100). out_act: 0 none, 1 sigmoid. */
IA_API int ia_ref_mlp_fwd(int64_t n, int n_layers, const int *dims, const float *const *W, const float *const *b,
                          int hidden_act, int out_act, const float *x, float *out)
{
    float buf0[256], buf1[256];
    for (int64_t i = 0; i < n; i++) {
        const float *in = x + i * dims[0];
        float *cur = buf0, *nxt = buf1;
        for (int k = 0; k < dims[0]; k++) cur[k] = in[k];
        for (int l = 0; l < n_layers; l++) {
            int di = dims[l], dout = dims[l + 1];
            for (int o = 0; o < dout; o++) {
                float acc = b[l][o];
                for (int k = 0; k < di; k++) acc = fmaf(W[l][o * di + k], cur[k], acc);
                if (l < n_layers - 1) acc = hidden_act == 0 ? fmaxf(acc, 0.0f) : softplus100(acc);
                else if (out_act == 1) acc = sigmoidf_(acc);
                nxt[o] = acc;
            }
            float *t = cur; cur = nxt; nxt = t;
        }
        for (int o = 0; o < dims[n_layers]; o++) out[i * dims[n_layers] + o] = cur[o];
    }
    return 0;
}

/* Laplace-CDF density + alpha: density.py:25-30, intrinsic_avatar.py:390-394 */
IA_API int ia_ref_laplace_alpha(int64_t n, const float *sdf, const float *dists, float beta, float *alpha)
{
    float inv = 1.0f / beta;
    for (int64_t i = 0; i < n; i++) {
        float s = sdf[i];
        float sg = (s > 0) - (s < 0);
        float dens = inv * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
        alpha[i] = 1.0f - expf(-dens * dists[i]);
    }
    return 0;
}

/* SDF field: VolumeSDF.forward (models/rf/geometry.py:124-172) with analytic normal.
 *   x' = (x - center)/scale + 0.5 ; h = [2x'-1, enc(x')*mask] (35) ; z = W1 h + b1 (64);
 *   a = softplus100(z); out = W2 a + b2 (13); sdf = out[0]; grad = d sdf / d x. */
IA_API int ia_ref_sdf_field(int64_t n, const float *x, const float *center, const float *scale,
                            const float *params, const float *level_mask /*[32]*/,
                            const float *W1 /*[64,35]*/, const float *b1, const float *W2 /*[13,64]*/,
                            const float *b2, float *sdf, float *grad /*[n,3] or NULL*/, float *feat /*[n,13] or NULL*/)
{
    const int L = 16, F = 2, H = 64, IN = 35, OUT = 13;
    float enc[32], denc[32 * 3], h[35], z[64], a[64];
    for (int64_t i = 0; i < n; i++) {
        float xp[3];
        for (int d = 0; d < 3; d++) xp[d] = (x[i * 3 + d] - center[d]) / scale[d] + 0.5f;
        ia_ref_hashgrid_fwd(1, xp, params, L, F, 19, 16, 1.447269237440378f, enc, grad ? denc : NULL);
        for (int d = 0; d < 3; d++) h[d] = xp[d] * 2.0f + -1.0f;
        for (int k = 0; k < 32; k++) h[3 + k] = enc[k] * level_mask[k];
        for (int o = 0; o < H; o++) {
            float acc = b1[o];
            for (int k = 0; k < IN; k++) acc = fmaf(W1[o * IN + k], h[k], acc);
            z[o] = acc;
            a[o] = softplus100(acc);
        }
        for (int o = 0; o < OUT; o++) {
            float acc = b2[o];
            for (int k = 0; k < H; k++) acc = fmaf(W2[o * H + k], a[k], acc);
            if (o == 0) sdf[i] = acc;
            if (feat) feat[i * OUT + o] = acc;
        }
        if (grad) {
            /* g_z = sigmoid(100 z) * W2[0,:]; g_h = W1^T g_z; grad = J_h^T g_h / scale */
            float gh[35];
            for (int k = 0; k < IN; k++) gh[k] = 0.0f;
            for (int o = 0; o < H; o++) {
                float gz = sigmoidf_(100.0f * z[o]) * W2[o];
                for (int k = 0; k < IN; k++) gh[k] = fmaf(W1[o * IN + k], gz, gh[k]);
            }
            for (int d = 0; d < 3; d++) {
                float g = 2.0f * gh[d];
                for (int k = 0; k < 32; k++) g = fmaf(gh[3 + k] * level_mask[k], denc[k * 3 + d], g);
                grad[i * 3 + d] = g / scale[d];
            }
        }
    }
    return 0;
}
