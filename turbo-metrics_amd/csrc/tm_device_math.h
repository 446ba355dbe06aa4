// tm_device_math.h -- gfx950 device-side scalar math for the SSIMULACRA2 path.
//
// The reference kernels call two libdevice routines whose bit-level behaviour is NVIDIA's:
//   __nv_cbrtf     in linear_to_xyb            (ssimulacra2-cuda-kernel/src/xyb.rs:44-46)
//   __nv_fast_powf in the BT.709 / sRGB EOTFs  (cuda-colorspace-kernel/src/lib.rs:228, srgb.rs:46)
// Here they are fixed sequences of IEEE-754 operations, deterministic and reproducible on any IEEE machine -- which is what
// lets the parity tests demand bit equality for every plane:
//   cbrt              f32 mul / sub / fma only, 21 operations, evaluated on pairs (v_pk_*_f32); <= 0.500002 ulp (11 of 25 M arguments not the nearest float)
//   BT.709 transfer   the reference's f32 base (v + a) / A, then its power from a table of 428 binary64 cubics (three v_fma_f64, one
//                     rounding): the correctly rounded value of the reference's expression for all but 117 of 15.4 M arguments
//                     (the reference's fast_powf: ~8 ulp)
//   pow (sRGB path of 16-bit / f32 RGB frames)  ~20 f64 operations, three 32-entry tables, one rounding to f32; <= 0.50001 ulp
// Cost matters: the ingest kernel is bound by its arithmetic, so none of the routines divides.
//
// Must be compiled with -ffp-contract=off: the fma() calls are the only fused operations.
#pragma once
#include "tm_platform.h" // the machine vocabulary (gfx950; host stand-ins for the no-GPU test tier)

namespace tmdev {

#define TM_MATH_INLINE __forceinline__

__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }

// ---- two-lane f32 vectors (tm_f2, f2_fma: tm_platform.h): gfx950 executes v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on two
// floats per lane in one VALU instruction, so everything below that can be said on pairs is said on pairs ------------
__device__ __forceinline__ tm_f2 f2_make(float a, float b) { tm_f2 v; v.x = a; v.y = b; return v; }
__device__ __forceinline__ tm_f2 f2_splat(float a) { return f2_make(a, a); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// Cube root, f32 only (no f64, no division): 21 IEEE operations.
//   r ~ a^(-1/3): exponent-trick seed (3.4 %), ONE sixth-order step r <- r (1 + e/3 + 2 e^2/9 + 14 e^3/81 + 35 e^4/243 + 91 e^5/729),
//                 e = 1 - a r^3 (up to 0.1 -> 1e-7);
//   y0 = (a r) r, then ONE Newton step on y whose residual a - y0^3 is formed exactly (s + se = y0^2 error-free, then two fma:
//   a - s y0 is exact inside the first, the second adds the -se y0 part), correction factor 1/(3 y^2) ~ r^2/3:
//   y = fma(res, c, y0) is the only rounding that matters -> |error| <= 0.500002 ulp.  Checked over EVERY float of [1, 8) (all
//   mantissas for each exponent residue mod 3; the sequence is exactly invariant under scaling by powers of 8) and of the
//   pixel-value range [0.0037, 1.004] (tools/check_cbrt.c): 11 of 25 165 824 arguments are not the correctly rounded value.
//   Valid for normal a in [2^-100, 2^100].  (Round 2's fifth-order step -- 20 operations -- left y0 up to 58 ulp out and the
//   Newton step's own second-order term at 2.6e-4 ulp: 933 arguments off; the first version -- two third-order steps and a
//   residual from four error-free operations, 27 operations -- missed the correctly rounded value for 2 arguments per three octaves.)
__device__ __forceinline__ tm_f2 cbrt_core2(tm_f2 a)
{
    // (floor(u / 3) through f64 -- three f64-rate instructions instead of v_mul_hi_u32 -- was measured: ingest 1.49 vs 1.44 ms, slower)
    tm_f2 r = f2_make(u2f(0x54a23400u - f2u(a.x) / 3u), u2f(0x54a23400u - f2u(a.y) / 3u));
    const tm_f2 one = f2_splat(1.0f), c13 = f2_splat(0x1.555556p-2f);
    {
        tm_f2 t = r * r;
        t = t * r;
        const tm_f2 e = f2_fma(-a, t, one);
        tm_f2 p = f2_fma(e, f2_splat(0x1.ff4c34p-4f), f2_splat(0x1.26fabcp-3f)); // 91/729, 35/243
        p = f2_fma(p, e, f2_splat(0x1.61f9aep-3f));                                // 14/81
        p = f2_fma(p, e, f2_splat(0x1.c71c72p-3f));                                // 2/9
        p = f2_fma(p, e, c13);                                                     // 1/3
        p = p * e;
        r = f2_fma(r, p, r);
    }
    const tm_f2 y0 = (a * r) * r;
    const tm_f2 s = y0 * y0, se = f2_fma(y0, y0, -s);
    tm_f2 res = f2_fma(-s, y0, a);
    res = f2_fma(-se, y0, res);
    const tm_f2 c = (r * r) * c13;
    return f2_fma(res, c, y0);
}

// any a: +0 / negative / NaN / inf are returned unchanged (the callers pass max(mixed, 0), xyb.rs:44); values outside
// [2^-100, 2^100] are brought into range by an exact power-of-eight scaling
__device__ TM_MATH_INLINE float cbrt_pos(float a)
{
    if (!(a > 0.0f) || !(a < __builtin_inff())) return a;
    float sc = 1.0f;
    if (a < 0x1p-100f) { a *= 0x1p96f; sc = 0x1p-32f; }
    else if (a > 0x1p100f) { a *= 0x1p-96f; sc = 0x1p32f; }
    return cbrt_core2(f2_splat(a)).x * sc;
}

// N cube roots in place; pixel data is always inside [2^-100, 2^100], so the pairs go through cbrt_core2 directly
// (same operations as cbrt_pos with sc = 1 -> same bits) and the general routine is only a fallback
// BOUNDED: the caller guarantees every value is inside that range (linear RGB in [0, 1] -> mixed in [0.0037, 1.004]), so
// the range test (two integer min / max per value) is not compiled at all
template <int N, bool BOUNDED = false> __device__ __forceinline__ void cbrt_pos_n(float (&v)[N])
{
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    if (!BOUNDED) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t u = f2u(v[i]);
            lo = u < lo ? u : lo;
            hi = u > hi ? u : hi;
        }
    }
    if (BOUNDED || (lo >= 0x0D800000u && hi <= 0x71800000u)) { // all in [2^-100, 2^100]: positive, normal, finite
#pragma unroll
        for (int i = 0; i + 1 < N; i += 2) {
            const tm_f2 y = cbrt_core2(f2_make(v[i], v[i + 1]));
            v[i] = y.x; v[i + 1] = y.y;
        }
        if (N & 1) v[N - 1] = cbrt_core2(f2_splat(v[N - 1])).x;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = cbrt_pos(v[i]);
    }
}

// x^y, finite x > 0.  tab: 96 doubles {rcp[32], nlog[32], exp2[32]} (tm_math_tables.inc), normally in LDS.
// ln x = e ln2 - ln(rcp_i) + log1p(m rcp_i - 1), |r| <= 2^-6, degree-6 series; exp z = 2^n 2^(j/32) exp(rr),
// |rr| <= ln2/64, degree-5 series.  ~20 f64 operations, no division.
__device__ TM_MATH_INLINE float pow_pos(float xf, double y, const double *__restrict__ tab)
{
    if (!(xf > 0.0f)) return xf != xf ? xf : 0.0f;
    const double x = (double)xf;
    const uint64_t b = d2u(x);
    const int e = (int)(b >> 52) - 1023;
    const int i = (int)(b >> 47) & 31;
    const double m = u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    const double r = __builtin_fma(m, tab[i], -1.0);
    double p = __builtin_fma(r, -0x1.5555555555555p-3, 0x1.999999999999ap-3);
    p = __builtin_fma(p, r, -0.25);
    p = __builtin_fma(p, r, 0x1.5555555555555p-2);
    p = __builtin_fma(p, r, -0.5);
    p = __builtin_fma(p, r, 1.0);
    const double ln2 = 0x1.62e42fefa39efp-1;
    const double lnx = __builtin_fma((double)e, ln2, tab[32 + i] + r * p);
    const double z = y * lnx;
    const double k = __builtin_rint(z * 0x1.71547652b82fep+5);
    const double rr = __builtin_fma(-k, 0x1.62e42fefa39efp-6, z);
    double q = __builtin_fma(rr, 0x1.1111111111111p-7, 0x1.5555555555555p-5);
    q = __builtin_fma(q, rr, 0x1.5555555555555p-3);
    q = __builtin_fma(q, rr, 0.5);
    q = __builtin_fma(q, rr, 1.0);
    q = __builtin_fma(q, rr, 1.0);
    const int ki = (int)k;
    const double res = q * tab[64 + (ki & 31)];
    return (float)u2d(d2u(res) + ((uint64_t)(long long)(ki >> 5) << 52));
}

// x / c for a positive constant c, as IEEE division rounds it, in 4 operations instead of the ~11 of the hardware division
// sequence: q1 = RN(x RN(1/c)), the exact remainder x - q1 c (one fma), one correction, the sign of x (so that -0 stays -0).
// For each c used below the three-operation core was compared with x / c over ALL 2^23 mantissas of x (tools/check_div_const.c:
// 0 mismatches; the uncorrected product alone is wrong for 3-10 % of them); the quotient's mantissa does not depend on the
// exponent of x, so the result holds wherever nothing under- or overflows: 2^-100 < |x| < 2^100 and x = +-0.  Callers: only
// the YUV transfer function, whose argument is a sum of two products of integer samples (|x| in {0} U [2^-40, 4]).
__device__ __forceinline__ float div_const(float x, float c, float rc)
{
    const float q1 = x * rc;
    const float r = __builtin_fmaf(-q1, c, x);
    return __builtin_copysignf(__builtin_fmaf(r, rc, q1), x);
}

// Layout of the math table buffer (tm_math_tables.inc, uploaded once per engine): 96 doubles of pow_pos {rcp[32], nlog[32], exp2[32]},
// then the binary64 transfer-function table: TM_EOTF64_SEGS records {c0, c1, c2, c3} of the cubic in t = 512 x - k (segment 512 is the
// constant 1 for bases >= 1; segments below 84 are never read).  32-byte records, 32-byte aligned.
#define TM_EOTF64_SEGS 513
#define TM_EOTF64_STRIDE 4
#define TM_TAB_POW_DOUBLES 96
#define TM_TAB_EOTF64 96 /* offset (doubles) of the transfer-function table */
#define TM_TAB_DOUBLES (TM_TAB_EOTF64 + TM_EOTF64_STRIDE * TM_EOTF64_SEGS)

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// x / c for x > 0 (no sign handling): the three-operation core of div_const
__device__ __forceinline__ float div_const_pos(float x, float c, float rc)
{
    const float q1 = x * rc;
    const float r = __builtin_fmaf(-q1, c, x);
    return __builtin_fmaf(r, rc, q1);
}

// s - floor(s) for s >= 0 (exact): v_fract_f32
__device__ __forceinline__ float fract_pos(float s) { return tm_fract_pos(s); }

// BT709::eotf, cuda-colorspace-kernel/src/lib.rs:221-236 (same body for both BT601 structs).  Power branch: the reference
// evaluates powf_fast((v + (ALPHA - 1)) / ALPHA, 1 / 0.45) (exp2(y log2 x), ~8 ulp).  Here the base x is formed with the
// reference's own two f32 operations -- the addition, then the IEEE quotient (div_const_pos by ALPHA / 512, which returns
// s = 512 x exactly: a power-of-two scaling of divisor and reciprocal scales every intermediate exactly) -- and x^(1/0.45f) is
// the cubic of one of 428 segments (t = s - k, k = floor(s), both exact in f32) with BINARY64 coefficients, three v_fma_f64, one
// rounding to binary32: the correctly rounded power of that base for all but 117 of the 15.4 M floats of [THRESHOLD, 1), never
// further than 0.500025 ulp (tests/test_oracle_pins.py scans them against long-double powl) -- practically the bits of "float64
// pow of the f32 base, rounded once", which the independent numpy twin evaluates (eotf = "exact").  History: round 2 fitted
// the real function of v in f32 (<= 0.68 ulp of THAT, up to 5 ulp from the expression as written, the 1080p NV12 golden 2.1e-2
// from the accurate evaluation); an f32 cubic on the f32 base (<= 0.69 ulp, 3 % of the arguments off by one ulp) still left
// it at 1.8e-2 -- that case moves by 7e-3 ... 4e-2 whenever 0.8 % of its samples move by ONE ulp (tools/score_conditioning.py)
// --; with this evaluation every golden sits at the level the 0.5003-ulp cube root alone costs (<= 2.4e-3).  x >= 1: the exact
// value is >= 1 and every caller clamps.  et64: the table (TM_EOTF64_SEGS records of 4 doubles), in LDS or global memory.
__device__ __forceinline__ float bt709_eotf(float v, const double *__restrict__ et64)
{
    const float THRESHOLD = 0.08124285829863521110029445797874f;
    const float BETA = 0.018053968510807f;
    const float ALPHA = 1.0f + 5.5f * BETA;
    if (v >= THRESHOLD) {
        const float s = div_const_pos(v + (ALPHA - 1.0f), ALPHA * 0.001953125f, 512.0f / ALPHA);
        if (s >= 512.0f) return 1.0f;
        const int k = (int)s;
        const double t = (double)fract_pos(s);
        const double *c = et64 + TM_EOTF64_STRIDE * k;
        double p = __builtin_fma(c[3], t, c[2]);
        p = __builtin_fma(p, t, c[1]);
        p = __builtin_fma(p, t, c[0]);
        return (float)p;
    }
    return div_const(v, 4.5f, 1.0f / 4.5f);
}

// (TM_WAVE_ANY -- "does any lane of the wave ...", a wave-uniform branch -- and TM_NO_IF_CONVERSION: tm_platform.h; every use below
// computes the same bits on either side of the branch)

// The power branch of bt709_eotf for TWO values at once (the ref and the dis sample of one pixel-channel: the side-packed ingest
// kernel), without a test: same operations as bt709_eotf on each component.  The base and its constant division (-> s = 512 x)
// run on the pair; s is clamped into [84, 512] so that the table index is always valid -- a base >= 1 lands on segment 512, the
// constant 1, an argument below THRESHOLD on some segment whose value the caller replaces (bt709_eotf_linear_fix).  Returns the
// UNCLAMPED values.
__device__ __forceinline__ tm_f2 bt709_power2(tm_f2 v, const double *__restrict__ et64)
{
    const float BETA = 0.018053968510807f;
    const float ALPHA = 1.0f + 5.5f * BETA;
    const float C = ALPHA * 0.001953125f, RC = 512.0f / ALPHA; // ALPHA / 512: the quotient is s = 512 x exactly
    const tm_f2 a = v + f2_splat(ALPHA - 1.0f);
    const tm_f2 q1 = a * f2_splat(RC);
    const tm_f2 rem = f2_fma(-q1, f2_splat(C), a);
    const tm_f2 s = f2_fma(rem, f2_splat(RC), q1); // 512 * RN((v + a) / A)
    const float s0 = fminf(fmaxf(s.x, 84.0f), 512.0f), s1 = fminf(fmaxf(s.y, 84.0f), 512.0f);
    const int k0 = (int)s0, k1 = (int)s1;
    const double t0 = (double)fract_pos(s0), t1 = (double)fract_pos(s1);
    // et64 here is the SPLIT copy the ingest kernel stages in LDS: {c0, c1}[513], then {c2, c3}[513] -- 16-byte records, so that two
    // lanes collide on a bank only when their segments are a multiple of 16 apart (32-byte records: 8; SQ_LDS_BANK_CONFLICT was two
    // thirds of the kernel's LDS cycles)
    const double *ca = et64 + 2 * k0, *cb = et64 + 2 * k1;
    const double *ca2 = ca + 2 * TM_EOTF64_SEGS, *cb2 = cb + 2 * TM_EOTF64_SEGS;
    double pa = __builtin_fma(ca2[1], t0, ca2[0]), pb = __builtin_fma(cb2[1], t1, cb2[0]);
    pa = __builtin_fma(pa, t0, ca[1]); pb = __builtin_fma(pb, t1, cb[1]);
    pa = __builtin_fma(pa, t0, ca[0]); pb = __builtin_fma(pb, t1, cb[0]);
    return f2_make((float)pa, (float)pb);
}

// the linear branch v / 4.5 for the components below THRESHOLD (negative arguments: the reference's quotient is negative and
// clamps to 0; so does the sign-free three-operation quotient used here -- its magnitude does not matter below 0)
__device__ __forceinline__ tm_f2 bt709_eotf_linear_fix(tm_f2 v, tm_f2 p)
{
    const float THRESHOLD = 0.08124285829863521110029445797874f;
    const float c = 4.5f, rc = 1.0f / 4.5f;
    const tm_f2 d1 = v * f2_splat(rc);
    const tm_f2 dr = f2_fma(-d1, f2_splat(c), v);
    const tm_f2 lin = f2_fma(dr, f2_splat(rc), d1);
    p.x = v.x >= THRESHOLD ? p.x : lin.x;
    p.y = v.y >= THRESHOLD ? p.y : lin.y;
    return p;
}
__device__ __forceinline__ tm_f2 clamp01_2(tm_f2 p) { return f2_make(clamp01(p.x), clamp01(p.y)); }

// srgb_inverse_oetf, cuda-colorspace-kernel/src/srgb.rs:40-48
__device__ __forceinline__ float srgb_inverse_oetf(float x, const double *__restrict__ tab)
{
    const float SRGB_ALPHA = 1.0550107f;
    const float SRGB_BETA = 0.0030412825f;
    if (x < 12.92f * SRGB_BETA) return x / 12.92f;
    return pow_pos((x + (SRGB_ALPHA - 1.0f)) / SRGB_ALPHA, (double)2.4f, tab);
}

// px_linear_rgb_to_positive_xyb, ssimulacra2-cuda-kernel/src/xyb.rs:42-79, for N pixels at once (the 3N cube roots
// are evaluated pairwise)
template <int N, bool BOUNDED = false>
__device__ __forceinline__ void linear_to_xyb_n(const float (&r)[N], const float (&g)[N], const float (&b)[N], float (&X)[N],
                                                float (&Y)[N], float (&B)[N])
{
    const float K_M02 = 0.078f, K_M00 = 0.30f, K_M01 = 1.0f - K_M02 - K_M00;
    const float K_M12 = 0.078f, K_M10 = 0.23f, K_M11 = 1.0f - K_M12 - K_M10;
    const float K_M20 = 0.24342269f, K_M21 = 0.20476745f, K_M22 = 1.0f - K_M20 - K_M21;
    const float K_B0 = 0.0037930734f;
    const float K_B0_ROOT = 0.1559542025327239180319220163705f;
    float m[3 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        m[3 * i + 0] = fmaxf(__builtin_fmaf(K_M00, r[i], __builtin_fmaf(K_M01, g[i], __builtin_fmaf(K_M02, b[i], K_B0))), 0.0f);
        m[3 * i + 1] = fmaxf(__builtin_fmaf(K_M10, r[i], __builtin_fmaf(K_M11, g[i], __builtin_fmaf(K_M12, b[i], K_B0))), 0.0f);
        m[3 * i + 2] = fmaxf(__builtin_fmaf(K_M20, r[i], __builtin_fmaf(K_M21, g[i], __builtin_fmaf(K_M22, b[i], K_B0))), 0.0f);
    }
    cbrt_pos_n<3 * N, BOUNDED>(m);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float rg = m[3 * i + 0] - K_B0_ROOT, gr = m[3 * i + 1] - K_B0_ROOT, bb = m[3 * i + 2] - K_B0_ROOT;
        const float x = 0.5f * (rg - gr);
        const float y = 0.5f * (rg + gr);
        X[i] = __builtin_fmaf(x, 14.0f, 0.42f);
        Y[i] = y + 0.01f;
        B[i] = bb - y + 0.55f;
    }
}

// the same for N pixels of BOTH sides at once (component x = ref, y = dis), linear RGB known to be inside [0, 1]: every cube root
// argument is >= K_B0 > 0, so the reference's max(mixed, 0) (xyb.rs:44) changes nothing and is not evaluated, and the pairs go
// through cbrt_core2 without a range test.  Same operations per component as linear_to_xyb_n -> same bits.
template <int N>
__device__ __forceinline__ void linear_to_xyb_sides(const tm_f2 (&r)[N], const tm_f2 (&g)[N], const tm_f2 (&b)[N], tm_f2 (&X)[N],
                                                    tm_f2 (&Y)[N], tm_f2 (&B)[N])
{
    const float K_M02 = 0.078f, K_M00 = 0.30f, K_M01 = 1.0f - K_M02 - K_M00;
    const float K_M12 = 0.078f, K_M10 = 0.23f, K_M11 = 1.0f - K_M12 - K_M10;
    const float K_M20 = 0.24342269f, K_M21 = 0.20476745f, K_M22 = 1.0f - K_M20 - K_M21;
    const float K_B0 = 0.0037930734f;
    const float K_B0_ROOT = 0.1559542025327239180319220163705f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const tm_f2 m0 = f2_fma(f2_splat(K_M00), r[i], f2_fma(f2_splat(K_M01), g[i], f2_fma(f2_splat(K_M02), b[i], f2_splat(K_B0))));
        const tm_f2 m1 = f2_fma(f2_splat(K_M10), r[i], f2_fma(f2_splat(K_M11), g[i], f2_fma(f2_splat(K_M12), b[i], f2_splat(K_B0))));
        const tm_f2 m2 = f2_fma(f2_splat(K_M20), r[i], f2_fma(f2_splat(K_M21), g[i], f2_fma(f2_splat(K_M22), b[i], f2_splat(K_B0))));
        const tm_f2 rg = cbrt_core2(m0) - f2_splat(K_B0_ROOT), gr = cbrt_core2(m1) - f2_splat(K_B0_ROOT), bb = cbrt_core2(m2) - f2_splat(K_B0_ROOT);
        const tm_f2 x = f2_splat(0.5f) * (rg - gr);
        const tm_f2 y = f2_splat(0.5f) * (rg + gr);
        X[i] = f2_fma(x, f2_splat(14.0f), f2_splat(0.42f));
        Y[i] = y + f2_splat(0.01f);
        B[i] = (bb - y) + f2_splat(0.55f);
    }
}

__device__ __forceinline__ void linear_to_xyb(float r, float g, float b, float &X, float &Y, float &B)
{
    const float ra[1] = {r}, ga[1] = {g}, ba[1] = {b};
    float xa[1], ya[1], za[1];
    linear_to_xyb_n<1>(ra, ga, ba, xa, ya, za);
    X = xa[0]; Y = ya[0]; B = za[0];
}

// One step of the three second-order sections of the truncated-cosine recursive Gaussian,
// ssimulacra2-cuda-kernel/src/blur.rs:112-131; constants = build.rs:28-145 for sigma 1.5.
struct Iir {
    float p1a, p1b, p1c, p2a, p2b, p2c;
};
__device__ __forceinline__ float iir_step(Iir &s, float sum)
{
    const float MUL_IN_1 = 0.055295236f, MUL_IN_3 = -0.058836687f, MUL_IN_5 = 0.012955819f;
    const float MUL_PREV_1 = 1.9021131f, MUL_PREV_3 = 1.1755705f, MUL_PREV_5 = 1.2246469e-16f;
    float o1 = sum * MUL_IN_1, o3 = sum * MUL_IN_3, o5 = sum * MUL_IN_5;
    o1 = __builtin_fmaf(-1.0f, s.p2a, o1);
    o3 = __builtin_fmaf(-1.0f, s.p2b, o3);
    o5 = __builtin_fmaf(-1.0f, s.p2c, o5);
    s.p2a = s.p1a; s.p2b = s.p1b; s.p2c = s.p1c;
    o1 = __builtin_fmaf(MUL_PREV_1, s.p1a, o1);
    o3 = __builtin_fmaf(MUL_PREV_3, s.p1b, o3);
    o5 = __builtin_fmaf(MUL_PREV_5, s.p1c, o5);
    s.p1a = o1; s.p1b = o3; s.p1c = o5;
    return (o1 + o3) + o5;
}

// the edge-difference half of compute_error_maps (error_maps.rs:45-59): needs mu1, mu2 only (d1 first, the two maps from it)
__device__ __forceinline__ float edge_d1(float source, float distorted, float mu1, float mu2)
{
    const float denom = 1.0f / (1.0f + fabsf(source - mu1));
    const float numer = 1.0f + fabsf(distorted - mu2);
    return __builtin_fmaf(numer, denom, -1.0f);
}
__device__ __forceinline__ void edge_from_d1(float d1, float &artifact, float &detail_loss)
{
    artifact = fmaxf(d1, 0.0f);
    detail_loss = fmaxf(-d1, 0.0f);
}
__device__ __forceinline__ void edge_maps(float source, float distorted, float mu1, float mu2, float &artifact,
                                          float &detail_loss)
{
    edge_from_d1(edge_d1(source, distorted, mu1, mu2), artifact, detail_loss);
}

// the ssim map of compute_error_maps (error_maps.rs:5-44): numerator and denominator, then the quotient
__device__ __forceinline__ void ssim_terms(float mu1, float mu2, float sigma11, float sigma22, float sigma12, float &num, float &den)
{
    const float C2 = 0.0009f;
    const float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
    const float mu_diff = mu1 - mu2;
    const float num_m = __builtin_fmaf(mu_diff, -mu_diff, 1.0f);
    const float num_s = __builtin_fmaf(2.0f, sigma12 - mu12, C2);
    den = (sigma11 - mu11) + (sigma22 - mu22) + C2;
    num = num_m * num_s;
}
__device__ __forceinline__ float ssim_from_terms(float num, float den) { return fmaxf(1.0f - num / den, 0.0f); }

// compute_error_maps, ssimulacra2-cuda-kernel/src/error_maps.rs:5-60
__device__ __forceinline__ void error_maps(float source, float distorted, float mu1, float mu2, float sigma11,
                                           float sigma22, float sigma12, float &ssim, float &artifact,
                                           float &detail_loss)
{
    float num, den;
    ssim_terms(mu1, mu2, sigma11, sigma22, sigma12, num, den);
    ssim = ssim_from_terms(num, den);
    edge_maps(source, distorted, mu1, mu2, artifact, detail_loss);
}

} // namespace tmdev
