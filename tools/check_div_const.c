/* Exhaustive check behind tmdev::div_const (turbo-metrics_amd/csrc/tm_device_math.h): for every constant the kernels divide
 * by on the YUV path, q = fma(fma(-q1, c, x), rc, q1) with q1 = x * rc, rc = RN(1/c) equals the IEEE quotient x / c for all
 * 2^23 mantissas of x.  gcc -O2 -ffp-contract=off tools/check_div_const.c -lm && ./a.out   (exit code 0 = all equal) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void)
{
    const float BETA = 0.018053968510807f, ALPHA = 1.0f + 5.5f * BETA;
    const float cs[] = {ALPHA, ALPHA * 0.001953125f /* bt709_eotf forms 512 x directly */, 4.5f};
    long total = 0;
    for (unsigned k = 0; k < sizeof cs / sizeof cs[0]; ++k) {
        volatile float c = cs[k];
        const float rc = 1.0f / c;
        long bad = 0, bad_plain = 0;
        for (uint32_t m = 0; m < (1u << 23); ++m) {
            const float x = u2f(0x3f800000u | m);
            const float q1 = x * rc, r = fmaf(-q1, c, x), q = fmaf(r, rc, q1), t = x / c;
            bad += q != t;
            bad_plain += q1 != t;
        }
        printf("c = %.9g: corrected sequence differs from x / c for %ld of 8388608 mantissas (x * (1/c) alone: %ld)\n", (double)c, bad, bad_plain);
        total += bad;
    }
    return total != 0;
}
