// t_advance.h -- exact closed form of k iterated float additions  t <- fl(t + s)  (s > 0, t >= 0).
//
// The marching recurrence t_{k+1} = t_k + dt is what makes sample positions bit-exact with the reference, but replaying
// it costs k dependent adds per element in the expansion pass.  Inside one binade [2^e, 2^(e+1)) every t is a multiple
// of u = ulp, so fl(t + s) = t + C*u with a CONSTANT integer C = round(s / u) -- except when s/u has fractional part
// exactly 1/2 (round-half-even then depends on the parity of t's mantissa; but every such result is even, so from the
// second step on the increment is constant again).  Hence: take two real steps inside a binade, read the increment off
// the bit patterns, jump to the end of the binade with one integer multiply-add, cross the boundary with a real add.
// O(#binades crossed) instead of O(k).  Compiles as C (gcc, for the CPU fuzz test) and as HIP device code; must be
// built without FMA contraction / fast-math like the rest of the bit-exact TUs.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define IA_ADV_FN __host__ __device__ __forceinline__
#else
#define IA_ADV_FN static inline
#endif

IA_ADV_FN uint32_t ia_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
IA_ADV_FN float ia_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

IA_ADV_FN float ia_advance(float t, float s, int k)
{
    if (!(t >= 0.0f) || !(s > 0.0f) || k <= 12) {                  // short runs: the plain recurrence is cheaper
        for (int i = 0; i < k; i++) t = t + s;
        return t;
    }
    while (k > 0) {
        const float b = t + s;
        k--;
        if (k == 0) return b;
        const float c = b + s;
        k--;
        const uint32_t ub = ia_f2u(b), uc = ia_f2u(c);
        if (k == 0) return c;
        const uint32_t eb = ub >> 23;
        // need b, c in one normal binade (t may still be in the binade below; the increment is read from b -> c)
        if (eb != (uc >> 23) || eb == 0u || eb >= 254u) { t = c; continue; }
        if (uc == ub) return c;                                     // s below half an ulp: t is stuck for good
        // b may have come out of an add that started in the binade below, so the tie-case parity argument covers c
        // onwards only: take the increment from one more in-binade add, c -> d.
        const float d = c + s;
        const uint32_t ud = ia_f2u(d);
        k--;
        if ((ud >> 23) != eb) { t = d; continue; }
        if (k == 0) return d;
        if (ud == uc) return d;                                     // stuck
        const uint32_t C2 = ud - uc;                                // mantissa increment per step inside this binade
        const uint32_t B = (eb + 1u) << 23;                         // first bit pattern of the next binade
        uint32_t jmax = (B - 1u - ud) / C2;                         // further steps that stay inside the binade
        const uint32_t j = jmax < (uint32_t)k ? jmax : (uint32_t)k;
        t = ia_u2f(ud + j * C2);
        k -= (int)j;
    }
    return t;
}
