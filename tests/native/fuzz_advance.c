#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "t_advance.h"
static uint64_t st = 88172645463325252ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static float urand(void) { return (float)((rnd() >> 40) / 16777216.0); }
int main(int argc, char** argv)
{
    long n = argc > 1 ? atol(argv[1]) : 2000000, bad = 0;
    for (long it = 0; it < n; it++) {
        float s, t;
        int mode = rnd() % 6;
        if (mode == 0) s = 4.3301f / 128.0f; else if (mode == 1) s = 1.5f / 63.0f;
        else if (mode == 2) { s = ia_u2f((ia_f2u(0.02f + urand() * 0.1f)) & ~((1u << (rnd() % 16)) - 1u)); }   // trailing zeros: tie cases
        else if (mode == 3) s = ldexpf(1.0f + (float)(rnd() % 8) / 8.0f, -(int)(rnd() % 12));                  // few mantissa bits
        else s = 1e-3f + urand() * 0.2f;
        int tm = rnd() % 4;
        if (tm == 0) t = 0.0f; else if (tm == 1) t = urand() * s; else if (tm == 2) t = urand() * 8.0f; else t = urand() * 200.0f;
        int k = rnd() % 400;
        float ref = t;
        for (int i = 0; i < k; i++) ref = ref + s;
        float got = ia_advance(t, s, k);
        if (ia_f2u(ref) != ia_f2u(got)) { if (bad < 10) printf("MISMATCH t=%a s=%a k=%d ref=%a got=%a\n", t, s, k, ref, got); bad++; }
    }
    printf("%ld cases, %ld mismatches\n", n, bad);
    return bad != 0;
}
