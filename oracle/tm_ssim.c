/*
 * oracle/tm_ssim.c -- TEST INFRASTRUCTURE ONLY (part of the parity oracle): SSIM and MS-SSIM of the u8-quantised
 * linear-RGB pair, the inputs of the reference's nppiSSIM_8u_C3R_Ctx / nppiWMSSSIM_8u_C3R_Ctx calls
 * (crates/turbo-metrics/src/lib.rs:296-340; wrappers crates/cudarse/cudarse-npp/src/image/ist.rs:106-179).
 *
 * PARITY UNPINNED.  NPP is closed source, is not under /root/reference, and no test, fixture or documented value in the
 * reference pins what these two calls return.  What follows is therefore BUILD-DEFINED: the published algorithms the
 * NPP entry points are named after, stated operation by operation so that the HIP kernels can be checked against it
 * bit for bit -- not a restatement of NPP.
 *   SSIM     Wang, Bovik, Sheikh, Simoncelli 2004: 11x11 Gaussian window (sigma 1.5, separable, normalised), windows
 *            that fit entirely inside the image, K1 = 0.01, K2 = 0.03, L = 255; mean over windows per channel; mean of
 *            the three channels.
 *   MS-SSIM  Wang, Simoncelli, Bovik 2003: five scales (2x2 box mean, decimate by 2, odd last row/column dropped),
 *            exponents 0.0448, 0.2856, 0.3001, 0.2363, 0.1333; product of the mean contrast-structure terms of scales
 *            1-4 and the mean SSIM of scale 5, per channel; mean of the three channels.  Needs min(w, h) >= 176.
 * Arithmetic: every per-pixel quantity in f32 with the fused operations written out, sums in f64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TMO_SSIM_SCALES 5
#define TMO_SSIM_TAPS 11

/* g[k] = exp(-(k-5)^2 / (2 * 1.5^2)) / sum, evaluated in f64 and rounded to f32 */
void tmo_ssim_window(float g[TMO_SSIM_TAPS])
{
    double v[TMO_SSIM_TAPS], s = 0.0;
    for (int k = 0; k < TMO_SSIM_TAPS; ++k) { const double d = (double)(k - 5); v[k] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); s += v[k]; }
    for (int k = 0; k < TMO_SSIM_TAPS; ++k) g[k] = (float)(v[k] / s);
}

/* sums over the (w-10) x (h-10) windows of one plane pair: out[0] = sum ssim, out[1] = sum cs
 * Four window means per pair: E[x], E[y], E[x^2 + y^2], E[xy] -- the two variances only ever appear as their sum
 * (sigma_x^2 + sigma_y^2 = E[x^2 + y^2] - mu_x^2 - mu_y^2), so they are filtered as one quantity, s = fma(x, x, y * y) per sample.
 * Filtering order: rows first (taps ascending, acc = fma(g[k], v, acc) from 0), then columns the same way. */
void tmo_ssim_plane_sums(const float *ref, const float *dis, int w, int h, double out[2])
{
    out[0] = out[1] = 0.0;
    if (w < TMO_SSIM_TAPS || h < TMO_SSIM_TAPS) return;
    float g[TMO_SSIM_TAPS];
    tmo_ssim_window(g);
    const int ow = w - 10, oh = h - 10;
    float *hx = malloc(sizeof(float) * 4 * (size_t)ow * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < ow; ++x) {
            float a[4] = {0, 0, 0, 0};
            for (int k = 0; k < TMO_SSIM_TAPS; ++k) {
                const float r = ref[(size_t)y * w + x + k], d = dis[(size_t)y * w + x + k];
                a[0] = fmaf(g[k], r, a[0]);
                a[1] = fmaf(g[k], d, a[1]);
                a[2] = fmaf(g[k], fmaf(r, r, d * d), a[2]);
                a[3] = fmaf(g[k], r * d, a[3]);
            }
            for (int q = 0; q < 4; ++q) hx[((size_t)q * h + y) * ow + x] = a[q];
        }
    const float C1 = 6.5025f, C2 = 58.5225f; /* (0.01*255)^2, (0.03*255)^2 */
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            float a[4] = {0, 0, 0, 0};
            for (int k = 0; k < TMO_SSIM_TAPS; ++k)
                for (int q = 0; q < 4; ++q) a[q] = fmaf(g[k], hx[((size_t)q * h + y + k) * ow + x], a[q]);
            const float mx = a[0], my = a[1];
            const float mxx = mx * mx, myy = my * my, mxy = mx * my, mm = mxx + myy;
            const float sv = a[2] - mm, sxy = a[3] - mxy; /* sigma_x^2 + sigma_y^2, sigma_xy */
            const float cs = fmaf(2.0f, sxy, C2) / (sv + C2);
            const float l = fmaf(2.0f, mxy, C1) / (mm + C1);
            out[0] += (double)(l * cs);
            out[1] += (double)cs;
        }
    free(hx);
}

/* 2x2 box mean, decimation by 2; an odd last row / column is dropped: ((a + b) + (c + d)) * 0.25 */
void tmo_ssim_downsample(const float *src, int w, int h, float *dst)
{
    const int dw = w / 2, dh = h / 2;
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const float *p = src + (size_t)(2 * y) * w + 2 * x;
            dst[(size_t)y * dw + x] = ((p[0] + p[1]) + (p[w] + p[w + 1])) * 0.25f;
        }
}

/* raw sums [channel 3][scale 5][ssim, cs] of the planar (3, h, w) u8 pair; scales that do not fit a window stay 0 */
void tmo_msssim_sums(const uint8_t *ref_q, const uint8_t *dis_q, int w, int h, double sums[30])
{
    memset(sums, 0, sizeof(double) * 30);
    float *a = malloc(sizeof(float) * (size_t)w * h), *b = malloc(sizeof(float) * (size_t)w * h);
    float *a2 = malloc(sizeof(float) * (size_t)w * h), *b2 = malloc(sizeof(float) * (size_t)w * h);
    for (int c = 0; c < 3; ++c) {
        for (size_t i = 0; i < (size_t)w * h; ++i) { a[i] = (float)ref_q[(size_t)c * w * h + i]; b[i] = (float)dis_q[(size_t)c * w * h + i]; }
        int sw = w, sh = h;
        float *pa = a, *pb = b, *qa = a2, *qb = b2;
        for (int s = 0; s < TMO_SSIM_SCALES; ++s) {
            tmo_ssim_plane_sums(pa, pb, sw, sh, sums + (c * TMO_SSIM_SCALES + s) * 2);
            tmo_ssim_downsample(pa, sw, sh, qa);
            tmo_ssim_downsample(pb, sw, sh, qb);
            float *t = pa; pa = qa; qa = t;
            t = pb; pb = qb; qb = t;
            sw /= 2; sh /= 2;
        }
    }
    free(a); free(b); free(a2); free(b2);
}

/* SSIM from the scale-0 sums: mean over windows per channel, mean of the channels, returned as the single Npp32f the
 * reference reads back (ist.rs:118,133) widened to f64 (lib.rs:355-357) */
double tmo_ssim_from_sums(const double sums[30], int w, int h)
{
    if (w < TMO_SSIM_TAPS || h < TMO_SSIM_TAPS) return NAN;
    const double n = (double)(w - 10) * (double)(h - 10);
    double acc = 0.0;
    for (int c = 0; c < 3; ++c) acc += sums[(c * TMO_SSIM_SCALES + 0) * 2] / n;
    return (double)(float)(acc / 3.0);
}

double tmo_msssim_from_sums(const double sums[30], int w, int h)
{
    static const double wt[TMO_SSIM_SCALES] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};
    if ((w >> 4) < TMO_SSIM_TAPS || (h >> 4) < TMO_SSIM_TAPS) return NAN;
    double acc = 0.0;
    for (int c = 0; c < 3; ++c) {
        double prod = 1.0;
        int sw = w, sh = h;
        for (int s = 0; s < TMO_SSIM_SCALES; ++s) {
            const double n = (double)(sw - 10) * (double)(sh - 10);
            const double v = sums[(c * TMO_SSIM_SCALES + s) * 2 + (s == TMO_SSIM_SCALES - 1 ? 0 : 1)] / n;
            prod *= pow(v > 0.0 ? v : 0.0, wt[s]); /* a negative mean contrast term has no real power: clamped */
            sw /= 2; sh /= 2;
        }
        acc += prod;
    }
    return (double)(float)(acc / 3.0);
}
