/* resample_harness.c -- host replay of the PRODUCT's re-sampling arithmetic (intrinsicavatar_amd/csrc/resample_math.h).
 *
 * resample.hip wraps the functions of that header in kernels: phase A = one lane per ray, phase B = one lane per output
 * element.  This file runs the same two phases as plain loops so that tests/test_resample_math_cpu.py can hold the product's
 * arithmetic to the golden vectors of the reference's kernels and to the oracle without a GPU.  Same entry-point shapes as
 * oracle/ia_oracle.c's ia_ref_ray_resampling*.  Built by the test with gcc -O2 -ffp-contract=off. */
#include <stdlib.h>
#include <string.h>
#include "../intrinsicavatar_amd/csrc/resample_math.h"

#define API __attribute__((visibility("default")))

API int rs_h_k1(int64_t n_rays, const int32_t* pinfo, const float* st, const float* en, const float* w, const float* sdf, int n,
                const int32_t* rpi, float* ts, float* offs, int64_t* surface_idx, int64_t* indices, int32_t* fg_counts, int32_t* bg_counts,
                int64_t n_in)
{
    float* utab = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    float* cdf = (float*)malloc(sizeof(float) * (size_t)(n_in + 1));
    float* cmax = (float*)malloc(sizeof(float) * (size_t)(n_in + 1));
    ia_rs_ray* hdr = (ia_rs_ray*)malloc(sizeof(ia_rs_ray) * (size_t)(n_rays + 1));
    ia_rs_fill_utab(n, 0, utab);
    for (int64_t r = 0; r < n_rays; r++) {                                 /* phase A */
        const int base = pinfo[2 * r], steps = pinfo[2 * r + 1];
        surface_idx[r] = -1;
        bg_counts[r] = 0;
        if (steps == 0) continue;
        int32_t surf, bg;
        ia_rs1_ray(steps, w + base, sdf + base, st + base, en + base, utab, n, cdf + base, cmax + base, fg_counts + base, &hdr[r], &surf, &bg);
        surface_idx[r] = surf >= 0 ? (int64_t)surf + base : -1;
        bg_counts[r] = bg;
    }
    for (int64_t r = 0; r < n_rays; r++) {                                 /* phase B (element order) */
        const int base = pinfo[2 * r], steps = pinfo[2 * r + 1];
        if (steps == 0) continue;
        const int rb = rpi[2 * r];
        for (int j = 0; j < n; j++) {
            int32_t k;
            ia_rs1_sample(j, steps, &hdr[r], st + base, en + base, cdf + base, cmax + base, utab, &ts[rb + j], &offs[rb + j], &k);
            indices[rb + j] = (int64_t)k + base;
        }
    }
    free(utab); free(cdf); free(cmax); free(hdr);
    return 0;
}

API int rs_h_k2(int64_t n_rays, const int32_t* pinfo, const float* vals, const uint8_t* il, const uint8_t* ir, const float* w, int n,
                const int32_t* rpi, float* ov, float* od, uint8_t* ol, uint8_t* orr, uint8_t* ors, uint8_t* fg, int64_t n_in)
{
    float* utab = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    float* cdf = (float*)malloc(sizeof(float) * (size_t)(n_in + 1));
    float* cmax = (float*)malloc(sizeof(float) * (size_t)(n_in + 1));
    int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_in + 1));
    ia_rs_ray* hdr = (ia_rs_ray*)malloc(sizeof(ia_rs_ray) * (size_t)(n_rays + 1));
    ia_rs_fill_utab(n, 0, utab);
    for (int64_t r = 0; r < n_rays; r++) {
        const int base = pinfo[2 * r], steps = pinfo[2 * r + 1];
        if (steps == 0) continue;
        ia_rs2_ray(steps, vals + base, il + base, ir + base, w + base, utab, n, cdf + base, cmax + base, first + base, &hdr[r]);
    }
    for (int64_t r = 0; r < n_rays; r++) {
        const int base = pinfo[2 * r], steps = pinfo[2 * r + 1];
        if (steps == 0) continue;
        const int rb = rpi[2 * r], cnt = rpi[2 * r + 1];
        for (int p = 0; p < cnt; p++) {
            const ia_rs2_edge e = ia_rs2_at(p, steps, &hdr[r], vals + base, il + base, ir + base, cdf + base, first + base, utab);
            float d = 0.0f;
            if (p + 1 < cnt) {
                const ia_rs2_edge f = ia_rs2_at(p + 1, steps, &hdr[r], vals + base, il + base, ir + base, cdf + base, first + base, utab);
                if (f.used) d = f.val - e.val;
            }
            ov[rb + p] = e.val; od[rb + p] = d; ol[rb + p] = e.left; orr[rb + p] = e.right; ors[rb + p] = e.resample; fg[rb + p] = e.used;
        }
    }
    free(utab); free(cdf); free(cmax); free(first); free(hdr);
    return 0;
}

API int rs_h_k34(int sdf_mode, int64_t n_rays, const int32_t* pinfo, const float* st, const float* en, const float* wa, const float* sdf,
                 int n, const int32_t* rpi, float* os, float* oe, uint8_t* fg, int64_t n_in)
{
    float* utab = (float*)malloc(sizeof(float) * (size_t)(n + 2));
    float* cdf = (float*)malloc(sizeof(float) * (size_t)(n_in + 1));
    float* cmax = (float*)malloc(sizeof(float) * (size_t)(n_in + 1));
    ia_rs_ray* hdr = (ia_rs_ray*)malloc(sizeof(ia_rs_ray) * (size_t)(n_rays + 1));
    ia_rs_fill_utab(n, 1, utab);
    for (int64_t r = 0; r < n_rays; r++) {
        const int base = pinfo[2 * r], steps = pinfo[2 * r + 1];
        if (steps == 0) continue;
        ia_rs34_ray(sdf_mode, steps, wa + base, sdf_mode ? sdf + base : (const float*)0, utab, n + 1, cdf + base, cmax + base, &hdr[r]);
    }
    for (int64_t r = 0; r < n_rays; r++) {
        const int base = pinfo[2 * r], steps = pinfo[2 * r + 1];
        if (steps == 0) continue;
        const int rb = rpi[2 * r];
        for (int q = 0; q < n; q++) {
            const int has_s = q < hdr[r].n_hit, has_e = q + 1 < hdr[r].n_hit;
            os[rb + q] = has_s ? ia_rs34_point(q, steps, &hdr[r], st + base, en + base, cdf + base, cmax + base, utab) : 0.0f;
            oe[rb + q] = has_e ? ia_rs34_point(q + 1, steps, &hdr[r], st + base, en + base, cdf + base, cmax + base, utab) : 0.0f;
            fg[rb + q] = (uint8_t)has_e;
        }
    }
    free(utab); free(cdf); free(cmax); free(hdr);
    return 0;
}

/* the register-resident form of K3 / K4 (n + 1 <= IA_RS_SMALL points per ray) */
API int rs_h_k34_small(int sdf_mode, int64_t n_rays, const int32_t* pinfo, const float* st, const float* en, const float* wa,
                       const float* sdf, int n, const int32_t* rpi, float* os, float* oe, uint8_t* fg)
{
    if (n + 1 > IA_RS_SMALL) return -1;
    const int bins = n + 1;
    const float du = (float)((1.0f - 1.0 / bins) / n);
    const float u0 = (float)(1.0 / (2 * bins));
    for (int64_t r = 0; r < n_rays; r++) {
        const int base = pinfo[2 * r], steps = pinfo[2 * r + 1];
        if (steps == 0) continue;
        float pts[IA_RS_SMALL];
        for (int j = 0; j < IA_RS_SMALL; j++) pts[j] = 0.0f;
        const int hit = ia_rs34_small(sdf_mode, bins, steps, wa + base, sdf_mode ? sdf + base : (const float*)0, st + base, en + base, du, u0, pts);
        const int rb = rpi[2 * r];
        for (int q = 0; q < n; q++) {
            os[rb + q] = q < hit ? pts[q] : 0.0f;
            oe[rb + q] = q + 1 < hit ? pts[q + 1] : 0.0f;
            fg[rb + q] = (uint8_t)(q + 1 < hit);
        }
    }
    return 0;
}
