// resample_math.h -- the arithmetic of the four importance re-sampling operators (lib/nerfacc/cuda/csrc/cdf.cu:10-696),
// restructured for a 64-wide machine.  Compiles as C (gcc: tests/resample_harness.c replays the kernels' phases on the host
// against the oracle) and as HIP device code (resample.hip wraps these functions in kernels).
//
// The reference walks each ray with one thread: a two-pointer merge of the ray's CDF (a serial fp32 running sum) with the
// sample positions u_j = u_0 + j du (an iterated fp32 sum), writing one output per step.  Only the two running sums have
// to be replayed in order to stay bit-exact; everything else is a function of (ray, j) alone:
//
//   * u_j depends on the per-ray sample count only, which is the same for every non-empty ray of a call: ONE table per launch
//     (ia_rs_fill_utab, n serial adds by one lane);
//   * phase A, one lane per ray: total weight, the CDF after each interval (cdf[]), its running maximum (cmax[]; equal to cdf[]
//     for non-negative weights -- the walk stops at the FIRST interval whose CDF exceeds u, which is what a search on the running
//     maximum finds whatever the signs), and the per-ray scalars (how many samples land before the CDF runs out, where the
//     zero-crossing clamp starts and which value it repeats).  Per-interval sample counts are differences of ranks in the u-table;
//   * phase B, one lane per OUTPUT element: rank -> interval by binary search in cmax[], then the reference's expression for
//     t, evaluated once.  Consecutive lanes write consecutive elements of every output array.
//
// The zero-crossing clamp of K1 ("ts[j] = ts[j-1]" once the interpolated SDF is negative, cdf.cu:86-103) is a chain through all
// later samples; it is resolved in closed form: inside the crossing interval the interpolated SDF is a non-increasing function of
// j (every operation of the expression is monotone in IEEE arithmetic), so the clamped samples are a suffix [j_clamp, n_fg) and
// all of them repeat the value of sample j_clamp - 1 (or the interval start of sample 0).  Phase A finds j_clamp by bisection.
//
// Must be built without FMA contraction / fast-math (expression order = oracle/ia_oracle.c).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define IA_RS_FN __host__ __device__ __forceinline__
#else
#include <math.h>
#define IA_RS_FN static inline
#endif

// per-ray record left by phase A for phase B
typedef struct {
    int32_t n_hit;      // samples that fall inside the CDF (K1: foreground samples; K2: inserted edges; K3/K4: emitted points)
    int32_t j_clamp;    // K1: first sample that repeats v_clamp (>= n_hit: none)
    float v_clamp;      // K1: the repeated value
    int32_t k_first;    // first interval of the walk (K4: the zero-crossing interval; otherwise 0)
} ia_rs_ray;

// ---- the sample positions -------------------------------------------------------------------------------------------------
// cdf.cu:53-57 (K1), :262-266 (K2): bins = n, du = (1 - 1/bins) / (n - 1); :440-444 (K3), :605-609 (K4): bins = n + 1, du = (1 - 1/bins) / n
IA_RS_FN void ia_rs_fill_utab(int n, int fine, float* utab)
{
    const int bins = fine ? n + 1 : n;
    const float du = (float)((1.0f - 1.0 / bins) / (fine ? n : n - 1));
    float u = (float)(1.0 / (2 * bins));
    for (int j = 0; j < bins; j++) { utab[j] = u; u += du; }
}

// number of table entries below v = index of the first sample whose u is not below v
IA_RS_FN int ia_rs_rank(const float* utab, int bins, float v)
{
    int lo = 0, hi = bins;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (utab[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// first interval of [lo, hi) whose running CDF maximum exceeds u (hi: none)
IA_RS_FN int ia_rs_interval(const float* cmax, int lo, int hi, float u)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cmax[mid] > u) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// position of the sample with CDF coordinate u inside interval k:  t = (u - cdf_lo) * ((b - a) / (cdf_hi - cdf_lo)) + a
// (cdf.cu:70-73); *offset gets the first product
IA_RS_FN float ia_rs_place(float u, float cdf_lo, float cdf_hi, float a, float b, float* offset)
{
    const float scaling = (b - a) / (cdf_hi - cdf_lo);
    const float off = (u - cdf_lo) * scaling;
    if (offset) *offset = off;
    return off + a;
}

// ---- K1: ray_resampling ---------------------------------------------------------------------------------------------------
// phase A for one ray.  w / sdf / st / en / cdf / cmax / fg_counts point at the ray's first interval.
IA_RS_FN void ia_rs1_ray(int steps, const float* w, const float* sdf, const float* st, const float* en, const float* utab, int bins,
                         float* cdf, float* cmax, int32_t* fg_counts, ia_rs_ray* out, int32_t* surface_local, int32_t* bg_count)
{
    float total = 0.0f;
    for (int k = 0; k < steps; k++) total += w[k];
    total += fmaxf(1.0f - total, 0.0f);
    // CDF, running maximum, per-interval sample counts; the zero-crossing interval and the ranks around it on the way
    int cross = -1, rank_before_cross = 0, rank_after_cross = 0;
    float run = 0.0f, top = -INFINITY;
    int rank_prev = 0;
    for (int k = 0; k < steps; k++) {
        const float share = w[k] / total;
        run = (k == 0) ? share : run + share;
        top = fmaxf(top, run);
        cdf[k] = run;
        cmax[k] = top;
        const int rank = ia_rs_rank(utab, bins, top);
        fg_counts[k] = rank - rank_prev;
        if (cross < 0 && k + 1 < steps && sdf[k] >= 0 && sdf[k + 1] < 0) { cross = k; rank_before_cross = rank_prev; rank_after_cross = rank; }
        rank_prev = rank;
    }
    const int n_hit = rank_prev;
    *bg_count = bins - n_hit;
    // the walk reports the crossing once it has LEFT that interval (cdf.cu:119-123): it leaves every interval when samples
    // remain after the CDF ran out, otherwise it ends in the interval of the last sample
    const int k_end = n_hit < bins ? steps - 1 : ia_rs_interval(cmax, 0, steps, utab[bins - 1]);
    *surface_local = (cross >= 0 && k_end > cross) ? cross : -1;
    int j_clamp = bins;
    float v_clamp = 0.0f;
    if (cross >= 0 && rank_before_cross < n_hit) {
        // samples [rank_before_cross, rank_after_cross) lie in the crossing interval: the first of them whose interpolated SDF
        // is not >= 0 (cdf.cu:88-97); all samples from rank_after_cross on are behind the surface
        const float lo_c = cross > 0 ? cdf[cross - 1] : 0.0f, hi_c = cdf[cross];
        const float a = st[cross], b = en[cross], s0 = sdf[cross], s1 = sdf[cross + 1];
        int lo = rank_before_cross, hi = rank_after_cross;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            float off;
            (void)ia_rs_place(utab[mid], lo_c, hi_c, a, b, &off);
            const float s = s0 + (s1 - s0) * (off / (b - a));
            if (s >= 0) lo = mid + 1; else hi = mid;
        }
        j_clamp = lo;
        if (j_clamp < n_hit) {
            if (j_clamp == 0) {
                v_clamp = st[ia_rs_interval(cmax, 0, steps, utab[0])];
            } else {
                const float u = utab[j_clamp - 1];
                const int k = ia_rs_interval(cmax, 0, steps, u);
                v_clamp = ia_rs_place(u, k > 0 ? cdf[k - 1] : 0.0f, cdf[k], st[k], en[k], (float*)0);
            }
        }
    }
    out->n_hit = n_hit;
    out->j_clamp = j_clamp;
    out->v_clamp = v_clamp;
    out->k_first = 0;
}

// phase B for sample j of a ray: (t, offset, interval index relative to the ray)
IA_RS_FN void ia_rs1_sample(int j, int steps, const ia_rs_ray* r, const float* st, const float* en, const float* cdf, const float* cmax,
                            const float* utab, float* t_out, float* off_out, int32_t* k_out)
{
    if (j >= r->n_hit) {                                    // behind the last interval (cdf.cu:131-146)
        const float off = 10000.f;
        *t_out = off + en[steps - 1];
        *off_out = off;
        *k_out = steps - 1;
        return;
    }
    const float u = utab[j];
    const int k = ia_rs_interval(cmax, 0, steps, u);
    float off;
    const float t = ia_rs_place(u, k > 0 ? cdf[k - 1] : 0.0f, cdf[k], st[k], en[k], &off);
    *t_out = j >= r->j_clamp ? r->v_clamp : t;
    *off_out = off;
    *k_out = k;
}

// ---- K2: ray_resampling_merge ---------------------------------------------------------------------------------------------
// The ray is an edge list; interval k = (vals[k], vals[k+1]) carries weight only where it is a real interval (is_left[k] &&
// is_right[k+1]); the output is the merge of the `steps` original edges with the inserted ones.  Original edge k lands at
// position k + (inserted edges in intervals < k), inserted edge j of interval k at position j + 1 + k (cdf.cu:283-331).
// phase A: cdf / cmax per interval [steps - 1], first[k] = rank of the first inserted edge not before original edge k [steps]
IA_RS_FN void ia_rs2_ray(int steps, const float* vals, const uint8_t* il, const uint8_t* ir, const float* w, const float* utab, int bins,
                         float* cdf, float* cmax, int32_t* first, ia_rs_ray* out)
{
    (void)vals;
    float total = 0.0f;
    for (int k = 0; k + 1 < steps; k++) total += (il[k] && ir[k + 1]) ? w[k] : 0.0f;
    total += fmaxf(1.0f - total, 0.0f);
    float run = 0.0f, top = -INFINITY;
    first[0] = 0;
    int rank = 0;
    for (int k = 0; k + 1 < steps; k++) {
        if (k == 0) run = w[0] / total;                     // interval 0 takes its weight whatever its flags say (cdf.cu:268)
        else if (il[k] && ir[k + 1]) run += w[k] / total;
        top = fmaxf(top, run);
        cdf[k] = run;
        cmax[k] = top;
        rank = ia_rs_rank(utab, bins, top);
        first[k + 1] = rank;
    }
    out->n_hit = rank;
    out->j_clamp = 0;
    out->v_clamp = 0.0f;
    out->k_first = 0;
}

typedef struct { float val; uint8_t used, left, right, resample; } ia_rs2_edge;

// phase B: output position p of a ray with `steps` original edges
IA_RS_FN ia_rs2_edge ia_rs2_at(int p, int steps, const ia_rs_ray* r, const float* vals, const uint8_t* il, const uint8_t* ir,
                               const float* cdf, const int32_t* first, const float* utab)
{
    ia_rs2_edge e;
    e.val = 0.0f; e.used = 0; e.left = 0; e.right = 0; e.resample = 0;
    if (p >= steps + r->n_hit) return e;                    // never reached: stays zero (cdf.cu:356-364 zero-initialises)
    // last original edge at or before p: positions k + first[k] are strictly increasing
    int lo = 0, hi = steps - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (mid + first[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int k = lo;
    e.used = 1;
    if (k + first[k] == p) {                                // the original edge itself
        e.val = vals[k];
        e.right = k > 0 ? ir[k] : 0;
        e.left = k == 0 ? 1 : (uint8_t)(k + 1 < steps && il[k] && ir[k + 1]);
    } else {                                                // inserted edge j of interval k
        const int j = p - k - 1;
        e.val = ia_rs_place(utab[j], k > 0 ? cdf[k - 1] : 0.0f, cdf[k], vals[k], vals[k + 1], (float*)0);
        e.left = 1; e.right = 1; e.resample = 1;
    }
    return e;
}

// ---- K3 / K4: ray_resampling_fine / ray_resampling_sdf_fine -----------------------------------------------------------------
// n + 1 points cut the CDF into n intervals; point j is the start of interval j and the end of interval j - 1 (cdf.cu:455-463).
// K3 normalises the weights like K1; K4 starts at the first zero crossing and builds transmittance weights from the alphas on
// the way, un-normalised (cdf.cu:573-600, 625-631).
IA_RS_FN void ia_rs34_ray(int sdf_mode, int steps, const float* wa, const float* sdf, const float* utab, int bins, float* cdf,
                          float* cmax, ia_rs_ray* out)
{
    int k0 = 0;
    float run = 0.0f, top = -INFINITY;
    out->j_clamp = 0;
    out->v_clamp = 0.0f;
    if (sdf_mode) {
        k0 = -1;
        for (int k = 0; k + 1 < steps; k++)
            if (sdf[k] >= 0 && sdf[k + 1] < 0) { k0 = k; break; }
        if (k0 < 0) { out->n_hit = 0; out->k_first = 0; return; }
        float trans = 1.0f;
        for (int k = k0; k < steps; k++) {
            const float a = wa[k];
            if (k == k0) run = a; else run += trans * a;
            trans *= (1.0f - a);
            top = fmaxf(top, run);
            cdf[k] = run;
            cmax[k] = top;
        }
    } else {
        float total = 0.0f;
        for (int k = 0; k < steps; k++) total += wa[k];
        total += fmaxf(1.0f - total, 0.0f);
        for (int k = 0; k < steps; k++) {
            const float share = wa[k] / total;
            run = (k == 0) ? share : run + share;
            top = fmaxf(top, run);
            cdf[k] = run;
            cmax[k] = top;
        }
    }
    out->n_hit = ia_rs_rank(utab, bins, top);
    out->k_first = k0;
}

// point j < n_hit of a ray
IA_RS_FN float ia_rs34_point(int j, int steps, const ia_rs_ray* r, const float* st, const float* en, const float* cdf, const float* cmax,
                             const float* utab)
{
    const float u = utab[j];
    const int k = ia_rs_interval(cmax, r->k_first, steps, u);
    return ia_rs_place(u, k > r->k_first ? cdf[k - 1] : 0.0f, cdf[k], st[k], en[k], (float*)0);
}

// ---- K3 / K4 with a handful of points per ray (the secondary rays' 4 intervals) ------------------------------------------------
// With n + 1 <= IA_RS_SMALL points the two phases collapse: one lane keeps the ray's points in registers, walks the intervals ONCE
// (the serial recurrence), and after each interval places the points whose u fell below the new CDF maximum -- their number is
// a rank in the (register-resident) u-table.  No tables leave the lane, and a ray's n outputs are written as one vector.
#define IA_RS_SMALL 9

// pts[0 .. bins) <- positions of the points that fall inside the CDF; returns how many do.  bins = n + 1 <= IA_RS_SMALL.
IA_RS_FN int ia_rs34_small(int sdf_mode, int bins, int steps, const float* wa, const float* sdf, const float* st, const float* en,
                           float du, float u0, float* pts)
{
    float u[IA_RS_SMALL];
    {
        float v = u0;
#pragma unroll
        for (int j = 0; j < IA_RS_SMALL; j++) { u[j] = v; v += du; }
    }
    int k0 = 0;
    float total = 1.0f;
    if (sdf_mode) {
        k0 = -1;
        float prev = sdf[0];
        for (int k = 0; k + 1 < steps; k++) {
            const float next = sdf[k + 1];
            if (prev >= 0 && next < 0) { k0 = k; break; }
            prev = next;
        }
        if (k0 < 0) return 0;
    } else {
        total = 0.0f;
        for (int k = 0; k < steps; k++) total += wa[k];
        total += fmaxf(1.0f - total, 0.0f);
    }
    float run = 0.0f, below = 0.0f, top = -INFINITY, trans = 1.0f;
    int placed = 0;
    for (int k = k0; k < steps && placed < bins; k++) {
        const float a = wa[k];
        if (sdf_mode) {
            if (k == k0) run = a; else run += trans * a;
            trans *= (1.0f - a);
        } else {
            const float share = a / total;
            run = (k == k0) ? share : run + share;
        }
        top = fmaxf(top, run);
        int rank = 0;
#pragma unroll
        for (int j = 0; j < IA_RS_SMALL; j++) rank += (j < bins && u[j] < top) ? 1 : 0;
        if (rank > placed) {
            const float a_k = st[k], b_k = en[k];
            const float scaling = (b_k - a_k) / (run - below);
#pragma unroll
            for (int j = 0; j < IA_RS_SMALL; j++)
                if (j >= placed && j < rank) pts[j] = (u[j] - below) * scaling + a_k;
            placed = rank;
        }
        below = run;
    }
    return placed;
}
