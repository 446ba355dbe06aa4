/*
 * oracle/tm_math.h -- TEST INFRASTRUCTURE ONLY (part of the parity oracle).
 *
 * Deterministic restatement of the two libdevice transcendental functions the
 * reference's kernels call:
 *   __nv_cbrtf      (crates/nvptx-std/src/math.rs:6-7, used by
 *                    crates/ssimulacra2-cuda-kernel/src/xyb.rs:44-46)
 *   __nv_fast_powf  (crates/nvptx-std/src/math.rs:14-15, used by the BT.709 EOTF
 *                    crates/cuda-colorspace-kernel/src/lib.rs:221-236 and the sRGB EOTF
 *                    crates/cuda-colorspace-kernel/src/srgb.rs:40-48)
 * libdevice is closed NVIDIA bitcode that is not under /root/reference, so its exact
 * rounding cannot be reproduced.  Both functions are restated as "the f32 value nearest the exact
 * result" computed through a fixed sequence of IEEE-754 operations (powf: binary64 +, -, *, fma, rint and
 * integer bit manipulation; cbrtf: binary32 *, -, fma).  The HIP kernels execute
 * the same sequence (turbo-metrics_amd/csrc/tm_device_math.h, written separately), and
 * because every step is an exactly specified IEEE operation the two agree bit for bit.
 * Accuracy (tests/test_oracle_pins.py, exhaustive scans): cbrtf <= 0.500002 ulp (11 of 25 M arguments not the nearest float); the BT.709 power branch is the correctly rounded
 * value of the reference's expression for all but 117 of 15.4 M arguments; powf <= 0.50001 ulp.  libdevice documents 1 ulp for
 * cbrtf and ~2 ulp + for fast_powf, i.e. the oracle sits inside the reference's own error band.
 *
 * Compile with -ffp-contract=off (the fma calls below are the only fused operations).
 */
#ifndef TM_ORACLE_MATH_H
#define TM_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint64_t tmo_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double tmo_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

static inline uint32_t tmo_f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float tmo_u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* cube root (the caller applies max(.,0) as xyb.rs:44 does), in 21 f32 operations:
 * r ~ a^(-1/3) from an exponent-trick seed (3.4 %) and one sixth-order step r <- r(1 + e/3 + 2e^2/9 + 14e^3/81 + 35e^4/243 +
 * 91e^5/729), e = 1 - a r^3; y0 = (a r) r; one Newton step on y with the residual a - y0^3 formed exactly (s + se = y0^2
 * error-free, two fma) and 1/(3y^2) ~ r^2/3.  The last fma is the only rounding that matters: |error| <= 0.500002 ulp, 11 of the
 * 25 M floats of [1, 8) and 33 of the 68 M of [0.0037, 1.004] are not the correctly rounded value (tools/check_cbrt.c,
 * tmo_cbrtf_scan); the product runs the same sequence on pairs.
 * +0, negatives, NaN, inf come back unchanged; arguments outside [2^-100, 2^100] are scaled by 8^(+-32) (exact). */
static inline float tmo_cbrtf(float a)
{
    if (!(a > 0.0f) || !(a < INFINITY)) return a;
    float sc = 1.0f;
    if (a < 0x1p-100f) { a *= 0x1p96f; sc = 0x1p-32f; }
    else if (a > 0x1p100f) { a *= 0x1p-96f; sc = 0x1p32f; }
    float r = tmo_u2f(0x54a23400u - tmo_f2u(a) / 3u);
    {
        float t = r * r;
        t = t * r;
        const float e = fmaf(-a, t, 1.0f);
        float p = fmaf(e, 0x1.ff4c34p-4f, 0x1.26fabcp-3f); /* 91/729, 35/243 */
        p = fmaf(p, e, 0x1.61f9aep-3f);                    /* 14/81 */
        p = fmaf(p, e, 0x1.c71c72p-3f);                    /* 2/9 */
        p = fmaf(p, e, 0x1.555556p-2f);                    /* 1/3 */
        p = p * e;
        r = fmaf(r, p, r);
    }
    const float y0 = (a * r) * r;
    const float s = y0 * y0, se = fmaf(y0, y0, -s);
    float res = fmaf(-s, y0, a);
    res = fmaf(-se, y0, res);
    const float c = (r * r) * 0x1.555556p-2f;
    return fmaf(res, c, y0) * sc;
}

#include "tm_math_tables.inc"
static const double tmo_pow_rcp[32] = {TM_POW_RCP};
static const double tmo_pow_nlog[32] = {TM_POW_NLOG};
static const double tmo_pow_exp2[32] = {TM_POW_EXP2};

/* x^y for finite x > 0; y is a compile-time exponent widened from f32.
 * ln x = e ln2 - ln(rcp_i) + log1p(m rcp_i - 1) with a 32-entry reciprocal table (|r| <= 2^-6, degree-6 series),
 * exp(z) = 2^n 2^(j/32) exp(rr) with |rr| <= ln2/64 (degree-5 series).  ~20 f64 operations, |error| ~ 1e-13. */
static inline float tmo_powf(float xf, double y)
{
    if (!(xf > 0.0f)) return xf != xf ? xf : 0.0f;
    const double x = (double)xf;
    const uint64_t b = tmo_d2u(x);
    const int e = (int)(b >> 52) - 1023;
    const int i = (int)(b >> 47) & 31;
    const double m = tmo_u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
    const double r = fma(m, tmo_pow_rcp[i], -1.0);
    double p = fma(r, -0x1.5555555555555p-3, 0x1.999999999999ap-3); /* -1/6, 1/5 */
    p = fma(p, r, -0.25);
    p = fma(p, r, 0x1.5555555555555p-2); /* 1/3 */
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    const double ln2 = 0x1.62e42fefa39efp-1;
    const double lnx = fma((double)e, ln2, tmo_pow_nlog[i] + r * p);
    const double z = y * lnx;
    const double k = rint(z * 0x1.71547652b82fep+5); /* 32 / ln 2 */
    const double rr = fma(-k, 0x1.62e42fefa39efp-6, z); /* ln2 / 32 */
    double q = fma(rr, 0x1.1111111111111p-7, 0x1.5555555555555p-5); /* 1/120, 1/24 */
    q = fma(q, rr, 0x1.5555555555555p-3); /* 1/6 */
    q = fma(q, rr, 0.5);
    q = fma(q, rr, 1.0);
    q = fma(q, rr, 1.0);
    const int ki = (int)k;
    const double res = q * tmo_pow_exp2[ki & 31];
    /* scale by 2^(ki>>5) through the exponent field (results stay far inside the normal range) */
    return (float)tmo_u2d(tmo_d2u(res) + ((uint64_t)(int64_t)(ki >> 5) << 52));
}

/* BT.709 transfer function, power branch (cuda-colorspace-kernel/src/lib.rs:228): powf_fast((v + (ALPHA - 1)) / ALPHA, 1 / 0.45),
 * in binary64: the reference's f32 base x (its two f32 operations: one addition, one IEEE division), then x^(1/0.45f) from one of 428 cubics in
 * t = 512 x - k with binary64 coefficients, three binary64 fma, ONE rounding to binary32 -- the correctly rounded value of the
 * reference's expression for all but ~1e-4 of the 15.4 M arguments (tmo_bt709_eotf_max_ulp2 counts them), i.e. practically the
 * bits of "float64 pow of the f32 base, rounded once", which is what the independent numpy twin evaluates
 * (oracle/twin_numpy.py eotf="exact").  The product runs the same sequence (tm_device_math.h bt709_eotf / bt709_power2; its
 * 4-operation constant division returns the IEEE quotient, tools/check_div_const.c).  x >= 1 (v >= 1, or a sum that rounds up to
 * ALPHA): the exact value is >= 1 and every caller clamps to 1. */
static const double tmo_eotf64_c[4 * 513] = {TM_EOTF64_C};
static inline float tmo_bt709_power(float v)
{
    const float BETA = 0.018053968510807f;
    const float ALPHA = 1.0f + 5.5f * BETA;
    const float x = (v + (ALPHA - 1.0f)) / ALPHA;
    const float s = x * 512.0f;
    if (s >= 512.0f) return 1.0f;
    const int k = (int)s;
    const double t = (double)(s - (float)k);
    const double *c = tmo_eotf64_c + 4 * k;
    double p = fma(c[3], t, c[2]);
    p = fma(p, t, c[1]);
    p = fma(p, t, c[0]);
    return (float)p;
}

#endif
