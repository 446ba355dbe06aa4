/* Exhaustive check of the product's / oracle's f32 cube root (tm_device_math.h cbrt_core2 == oracle/tm_math.h tmo_cbrtf):
 * every float of [1, 8) (all mantissas for each exponent residue mod 3; the sequence is exactly scale-invariant under powers of
 * 8) and every float of the pixel-value range [0.0037, 1.004] against long-double cbrtl.  Prints the largest error in ulps of
 * the exact result and the number of results that are not the correctly rounded one.
 *   gcc -O2 -ffp-contract=off -I oracle tools/check_cbrt.c -lm -o /tmp/check_cbrt && /tmp/check_cbrt */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "tm_math.h" /* oracle/tm_math.h: the CPU statement of the same sequence */

static void scan(float lo, float hi)
{
    double worst = 0.0;
    float at = 0.0f;
    long n = 0, not_rn = 0;
    for (float a = lo; a < hi; a = nextafterf(a, 1e30f), ++n) {
        const float got = tmo_cbrtf(a);
        const long double exact = cbrtl((long double)a);
        int e;
        (void)frexpl(exact, &e);
        const double err = (double)(fabsl((long double)got - exact) / ldexpl(1.0L, e - 24));
        if (err > worst) { worst = err; at = a; }
        if (got != (float)exact) ++not_rn; /* (float) of the long double = round to nearest (no double rounding issue at 64-bit mantissa for these) */
    }
    printf("[%g, %g): %ld floats, max error %.6f ulp at %a, %ld not correctly rounded\n", (double)lo, (double)hi, n, worst, (double)at, not_rn);
}

int main(void)
{
    scan(1.0f, 8.0f);
    scan(0.0037f, 1.004f);
    return 0;
}
