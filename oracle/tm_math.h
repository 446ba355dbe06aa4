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
 * rounding cannot be reproduced.  Both functions are restated as "the correctly rounded
 * f32 result" computed through a fixed sequence of IEEE-754 binary64 operations
 * (+, -, *, /, fma, rint and integer bit manipulation only).  The HIP kernels execute
 * the same sequence (turbo-metrics_amd/csrc/tm_device_math.h, written separately), and
 * because every step is an exactly specified IEEE operation the two agree bit for bit.
 * Accuracy: |error| < 0.5000001 ulp(f32) (tests/test_oracle_math.py); libdevice documents
 * 1 ulp for cbrtf and ~2 ulp + for fast_powf, i.e. the oracle sits inside the reference's
 * own error band.
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

/* cube root of a >= 0 (the caller applies max(.,0) as xyb.rs:44 does). */
static inline float tmo_cbrtf(float a)
{
    if (!(a > 0.0f)) return a; /* +0 -> +0; NaN -> NaN */
    const double x = (double)a;
    /* r ~= x^(-1/3): exponent trick on the high word, |rel err| < 3.5 % */
    const uint32_t hi = (uint32_t)(tmo_d2u(x) >> 32);
    double r = tmo_u2d((uint64_t)(0x553EF000u - hi / 3u) << 32);
    const double third = 0x1.5555555555555p-2;
    for (int i = 0; i < 4; ++i) { /* Newton on r^-3 = x, quadratic, division free */
        const double r3 = (r * r) * r;
        const double e = fma(-x, r3, 1.0);
        r = fma(r * third, e, r);
    }
    return (float)(x * (r * r));
}

/* x^y for finite x > 0; y is a compile-time exponent widened from f32. */
static inline float tmo_powf(float xf, double y)
{
    if (!(xf > 0.0f)) return xf != xf ? xf : 0.0f;
    const double x = (double)xf;
    const uint64_t b = tmo_d2u(x);
    int e = (int)(b >> 52) - 1023;
    double m = tmo_u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }                      /* (sqrt.5, sqrt2] */
    const double t = (m - 1.0) / (m + 1.0);
    const double t2 = t * t;
    /* ln m = 2 atanh t = 2 t (1 + t2/3 + t2^2/5 + ... + t2^10/21) */
    double s = 0x1.8618618618618p-5;        /* 1/21 */
    s = fma(s, t2, 0x1.af286bca1af28p-5);   /* 1/19 */
    s = fma(s, t2, 0x1.e1e1e1e1e1e1ep-5);   /* 1/17 */
    s = fma(s, t2, 0x1.1111111111111p-4);   /* 1/15 */
    s = fma(s, t2, 0x1.3b13b13b13b14p-4);   /* 1/13 */
    s = fma(s, t2, 0x1.745d1745d1746p-4);   /* 1/11 */
    s = fma(s, t2, 0x1.c71c71c71c71cp-4);   /* 1/9  */
    s = fma(s, t2, 0x1.2492492492492p-3);   /* 1/7  */
    s = fma(s, t2, 0x1.999999999999ap-3);   /* 1/5  */
    s = fma(s, t2, 0x1.5555555555555p-2);   /* 1/3  */
    s = fma(s, t2, 1.0);
    const double ln2 = 0x1.62e42fefa39efp-1;
    const double lnx = fma((double)e, ln2, (2.0 * t) * s);
    const double z = y * lnx;
    const double n = rint(z * 0x1.71547652b82fep+0); /* log2(e) */
    const double r = fma(-n, ln2, z);                /* |r| <= ~0.347 */
    /* exp r, Taylor to r^13 */
    double p = 0x1.6124613a86d09p-33;       /* 1/13! */
    p = fma(p, r, 0x1.1eed8eff8d898p-29);   /* 1/12! */
    p = fma(p, r, 0x1.ae64567f544e4p-26);   /* 1/11! */
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);   /* 1/10! */
    p = fma(p, r, 0x1.71de3a556c734p-19);   /* 1/9!  */
    p = fma(p, r, 0x1.a01a01a01a01ap-16);   /* 1/8!  */
    p = fma(p, r, 0x1.a01a01a01a01ap-13);   /* 1/7!  */
    p = fma(p, r, 0x1.6c16c16c16c17p-10);   /* 1/6!  */
    p = fma(p, r, 0x1.1111111111111p-7);    /* 1/5!  */
    p = fma(p, r, 0x1.5555555555555p-5);    /* 1/4!  */
    p = fma(p, r, 0x1.5555555555555p-3);    /* 1/3!  */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    /* scale by 2^n through the exponent field (results stay far inside the normal range) */
    const int64_t ni = (int64_t)n;
    return (float)tmo_u2d(tmo_d2u(p) + ((uint64_t)ni << 52));
}

#endif
