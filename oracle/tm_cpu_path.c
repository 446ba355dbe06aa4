/*
 * oracle/tm_cpu_path.c -- TEST INFRASTRUCTURE ONLY.
 *
 * C restatement of the reference's *CPU* path, crates/ssimulacra2-cuda/examples/cpu.rs (itself extracted
 * from rust-av/ssimulacra2): BASELINE.json configs[0] ("Single 1080p PNG pair, SSIMULACRA2 on reference Rust
 * CPU path") and the "restated reference CPU path" baseline timed beside the GPU numbers.  The Rust original
 * cannot be built here (no rustc/cargo).  Single-threaded like the original (cpu.rs:955 is the
 * non-rayon horizontal pass).
 *
 * This path differs from the GPU arithmetic restated in tm_oracle.c: it stops at scales
 * smaller than 8 px (cpu.rs:359-361), blurs horizontally then vertically (cpu.rs:921-928), its vertical
 * pass uses o = fma(sum, n2, -fma(prev, d1, prev2)) (cpu.rs:1093-1099), it evaluates the maps in f64
 * (cpu.rs:627-631, 659-673) and uses the platform cbrtf (Rust f32::cbrt -> libm).  The two paths agree to
 * ~1e-3 on the score, not bit for bit; the reference's own check allows 0.25 (examples/compare.rs:72).
 * Planar layout (3 planes of w*h); every operation is per sample so packed vs planar changes nothing.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tm_oracle_tables.inc"

static const double k_w[108] = {TM_SSIMU2_WEIGHTS};
static const uint32_t k_lut_bits[256] = {TM_SRGB_LUT_BITS};

int tmo_cpu_path_present(void) { return 1; }

/* consts, cpu.rs:931-948 */
static const float RG_MUL_IN[3] = {0.055295236f, -0.058836687f, 0.012955819f};
static const float RG_MUL_PREV[3] = {1.9021131f, 1.1755705f, 0.00000000000000012246469f};
static const float RG_MUL_PREV2 = -1.0f;
static const float RG_VERT_MUL_IN[3] = {0.055295236f, -0.058836687f, 0.012955819f};
static const float RG_VERT_MUL_PREV[3] = {-1.9021131f, -1.1755705f, -0.00000000000000012246469f};

/* RecursiveGaussian::horizontal_row, cpu.rs:967-1022 */
static void horizontal_row(const float *in, float *out, int width)
{
    const int N = 5;
    float p1 = 0, p3 = 0, p5 = 0, q1 = 0, q3 = 0, q5 = 0;
    for (int n = -N + 1; n < width; ++n) {
        const int left = n - N - 1, right = n + N - 1;
        const float lv = left >= 0 ? in[left] : 0.0f;
        const float rv = right < width ? in[right] : 0.0f;
        const float sum = lv + rv;
        float o1 = sum * RG_MUL_IN[0], o3 = sum * RG_MUL_IN[1], o5 = sum * RG_MUL_IN[2];
        o1 = fmaf(RG_MUL_PREV2, q1, o1); o3 = fmaf(RG_MUL_PREV2, q3, o3); o5 = fmaf(RG_MUL_PREV2, q5, o5);
        q1 = p1; q3 = p3; q5 = p5;
        o1 = fmaf(RG_MUL_PREV[0], p1, o1); o3 = fmaf(RG_MUL_PREV[1], p3, o3); o5 = fmaf(RG_MUL_PREV[2], p5, o5);
        p1 = o1; p3 = o3; p5 = o5;
        if (n >= 0) out[n] = o1 + o3 + o5;
    }
}

/* RecursiveGaussian::vertical_pass, cpu.rs:1054-1115 (column chunking :1024-1052 does not change arithmetic) */
static void vertical_pass(const float *in, float *out, int width, int height)
{
    const int N = 5;
    float *prev = calloc((size_t)3 * width, sizeof(float)), *prev2 = calloc((size_t)3 * width, sizeof(float));
    float *cur = calloc((size_t)3 * width, sizeof(float));
    for (int n = -N + 1; n < height; ++n) {
        const int top = n - N - 1, bottom = n + N - 1;
        for (int i = 0; i < width; ++i) {
            const float tv = top >= 0 ? in[(size_t)top * width + i] : 0.0f;
            const float bv = bottom < height ? in[(size_t)bottom * width + i] : 0.0f;
            const float sum = tv + bv;
            float o[3];
            for (int k = 0; k < 3; ++k) {
                const float t = fmaf(prev[k * width + i], RG_VERT_MUL_PREV[k], prev2[k * width + i]);
                o[k] = fmaf(sum, RG_VERT_MUL_IN[k], -t);
                cur[k * width + i] = o[k];
            }
            if (n >= 0) out[(size_t)n * width + i] = o[0] + o[1] + o[2];
        }
        memcpy(prev2, prev, (size_t)3 * width * sizeof(float));
        memcpy(prev, cur, (size_t)3 * width * sizeof(float));
    }
    free(prev); free(prev2); free(cur);
}

/* Blur::blur_plane, cpu.rs:921-928 */
static void blur_plane(const float *plane, float *temp, float *out, int w, int h)
{
    for (int y = 0; y < h; ++y) horizontal_row(plane + (size_t)y * w, temp + (size_t)y * w, w);
    vertical_pass(temp, out, w, h);
}

/* px_linear_rgb_to_xyb + opsin_absorbance, cpu.rs:460-507 (platform cbrtf, as Rust's f32::cbrt) */
static void linear_to_xyb(const float *lin, size_t n, float *xyb)
{
    const float K_M02 = 0.078f, K_M00 = 0.30f, K_M01 = 1.0f - K_M02 - K_M00;
    const float K_M12 = 0.078f, K_M10 = 0.23f, K_M11 = 1.0f - K_M12 - K_M10;
    const float K_M20 = 0.24342269f, K_M21 = 0.20476745f, K_M22 = 1.0f - K_M20 - K_M21;
    const float K_B0 = 0.0037930734f, K_B0_ROOT = 0.1559542025327239180319220163705f;
    for (size_t i = 0; i < n; ++i) {
        const float r = lin[i], g = lin[n + i], b = lin[2 * n + i];
        float rg = fmaf(K_M00, r, fmaf(K_M01, g, fmaf(K_M02, b, K_B0)));
        float gr = fmaf(K_M10, r, fmaf(K_M11, g, fmaf(K_M12, b, K_B0)));
        float bb = fmaf(K_M20, r, fmaf(K_M21, g, fmaf(K_M22, b, K_B0)));
        rg = cbrtf(fmaxf(rg, 0.0f)) - K_B0_ROOT;
        gr = cbrtf(fmaxf(gr, 0.0f)) - K_B0_ROOT;
        bb = cbrtf(fmaxf(bb, 0.0f)) - K_B0_ROOT;
        const float x = 0.5f * (rg - gr), y = 0.5f * (rg + gr);
        xyb[i] = fmaf(x, 14.0f, 0.42f);
        xyb[n + i] = y + 0.01f;
        xyb[2 * n + i] = bb - y + 0.55f;
    }
}

/* downscale_by_2, cpu.rs:545-579 */
static void downscale(const float *src, int sw, int sh, float *dst)
{
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    for (int oy = 0; oy < dh; ++oy)
        for (int ox = 0; ox < dw; ++ox) {
            float sum = 0.0f;
            for (int iy = 0; iy < 2; ++iy)
                for (int ix = 0; ix < 2; ++ix) {
                    int x = ox * 2 + ix; if (x > sw - 1) x = sw - 1;
                    int y = oy * 2 + iy; if (y > sh - 1) y = sh - 1;
                    sum += src[(size_t)y * sw + x];
                }
            dst[(size_t)oy * dw + ox] = sum * 0.25f;
        }
}

static inline double pow4(double d) { const double d2 = d * d; return d2 * d2; } /* f64::powi(4) */

/* compute_frame_ssimulacra2, cpu.rs:342-410 + ssim_map :581-638 + edge_diff_map :640-683 + Msssim::score :728-871.
 * ref_lin / dis_lin: planar linear RGB. */
double tmo_cpu_path_score_linear(const float *ref_lin, const float *dis_lin, int w0, int h0)
{
    int w = w0, h = h0;
    const size_t n0 = (size_t)w * h;
    float *img[2], *nxt[2], *xyb[2];
    for (int i = 0; i < 2; ++i) { img[i] = malloc(3 * n0 * 4); nxt[i] = malloc(3 * n0 * 4); xyb[i] = malloc(3 * n0 * 4); }
    memcpy(img[0], ref_lin, 3 * n0 * 4); memcpy(img[1], dis_lin, 3 * n0 * 4);
    float *mul = malloc(n0 * 4), *temp = malloc(n0 * 4);
    float *s11 = malloc(n0 * 4), *s22 = malloc(n0 * 4), *s12 = malloc(n0 * 4), *mu1 = malloc(n0 * 4), *mu2 = malloc(n0 * 4);
    double avg_ssim[6][6], avg_edge[6][12];
    int nscales = 0;
    for (int scale = 0; scale < 6; ++scale) {
        if (w < 8 || h < 8) break;
        if (scale > 0) {
            const int pw = w, ph = h;
            w = (pw + 1) / 2; h = (ph + 1) / 2;
            for (int i = 0; i < 2; ++i) {
                for (int c = 0; c < 3; ++c) downscale(img[i] + (size_t)c * pw * ph, pw, ph, nxt[i] + (size_t)c * w * h);
                float *t = img[i]; img[i] = nxt[i]; nxt[i] = t;
            }
        }
        const size_t n = (size_t)w * h;
        linear_to_xyb(img[0], n, xyb[0]);
        linear_to_xyb(img[1], n, xyb[1]);
        const double opp = 1.0 / (double)(w * h);
        for (int c = 0; c < 3; ++c) {
            const float *a = xyb[0] + c * n, *b = xyb[1] + c * n;
            for (size_t i = 0; i < n; ++i) mul[i] = a[i] * a[i];
            blur_plane(mul, temp, s11, w, h);
            for (size_t i = 0; i < n; ++i) mul[i] = b[i] * b[i];
            blur_plane(mul, temp, s22, w, h);
            for (size_t i = 0; i < n; ++i) mul[i] = a[i] * b[i];
            blur_plane(mul, temp, s12, w, h);
            blur_plane(a, temp, mu1, w, h);
            blur_plane(b, temp, mu2, w, h);
            double ss0 = 0, ss1 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
            const float C2 = 0.0009f;
            for (size_t i = 0; i < n; ++i) {
                const float m1 = mu1[i], m2 = mu2[i];
                const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2, md = m1 - m2;
                const float num_m = fmaf(md, -md, 1.0f);
                const float num_s = fmaf(2.0f, s12[i] - m12, C2);
                const float denom_s = (s11[i] - m11) + (s22[i] - m22) + C2;
                double d = 1.0 - (double)((num_m * num_s) / denom_s);
                d = fmax(d, 0.0);
                ss0 += d; ss1 += pow4(d);
                const double d1 = (1.0 + (double)fabsf(b[i] - m2)) / (1.0 + (double)fabsf(a[i] - m1)) - 1.0;
                const double art = fmax(d1, 0.0), det = fmax(-d1, 0.0);
                e0 += art; e1 += pow4(art); e2 += det; e3 += pow4(det);
            }
            avg_ssim[nscales][c * 2] = opp * ss0;
            avg_ssim[nscales][c * 2 + 1] = sqrt(sqrt(opp * ss1));
            avg_edge[nscales][c * 4] = opp * e0;
            avg_edge[nscales][c * 4 + 1] = sqrt(sqrt(opp * e1));
            avg_edge[nscales][c * 4 + 2] = opp * e2;
            avg_edge[nscales][c * 4 + 3] = sqrt(sqrt(opp * e3));
        }
        ++nscales;
    }
    double ssim = 0.0;
    int i = 0; /* NB: indexes the weight table by the scales that EXIST, cpu.rs:843-853 */
    for (int c = 0; c < 3; ++c)
        for (int s = 0; s < nscales; ++s)
            for (int k = 0; k < 2; ++k) {
                ssim = fma(k_w[i], fabs(avg_ssim[s][c * 2 + k]), ssim); ++i;
                ssim = fma(k_w[i], fabs(avg_edge[s][c * 4 + k]), ssim); ++i;
                ssim = fma(k_w[i], fabs(avg_edge[s][c * 4 + k + 2]), ssim); ++i;
            }
    ssim *= 0.9562382616834844;
    ssim = fma(6.248496625763138e-5 * ssim * ssim, ssim, fma(2.326765642916932, ssim, -0.020884521182843837 * ssim * ssim));
    if (ssim > 0.0) ssim = fma(pow(ssim, 0.6276336467831387), -10.0, 100.0);
    else ssim = 100.0;
    for (int k = 0; k < 2; ++k) { free(img[k]); free(nxt[k]); free(xyb[k]); }
    free(mul); free(temp); free(s11); free(s22); free(s12); free(mu1); free(mu2);
    return ssim;
}

/* CpuImg::from_srgb (cpu.rs:280-296) + compute_frame_ssimulacra2: packed sRGB u8 in, score out */
double tmo_cpu_path_score_srgb8(const uint8_t *ref, const uint8_t *dis, int w, int h)
{
    const size_t n = (size_t)w * h;
    float *a = malloc(3 * n * 4), *b = malloc(3 * n * 4);
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            float f; uint32_t u;
            u = k_lut_bits[ref[3 * i + c]]; memcpy(&f, &u, 4); a[c * n + i] = f;
            u = k_lut_bits[dis[3 * i + c]]; memcpy(&f, &u, 4); b[c * n + i] = f;
        }
    const double s = tmo_cpu_path_score_linear(a, b, w, h);
    free(a); free(b);
    return s;
}
