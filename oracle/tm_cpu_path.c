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
/* workspace of one evaluation: 25 planes of w0 * h0 floats (the three image buffers of each side + seven planes of scratch).
 * tmo_cpu_path_score_linear allocates one per call, like the original allocates per frame; the threaded runner below (the all-core
 * CPU baseline of bench.py) gives every worker ONE for its whole run -- 256 threads that each mmap and fault in 200 MB per pair
 * measure the kernel's page allocator, not the path. */
size_t tmo_cpu_path_ws_bytes(int w0, int h0) { return (size_t)25 * (size_t)w0 * (size_t)h0 * sizeof(float); }

double tmo_cpu_path_score_linear_ws(const float *ref_lin, const float *dis_lin, int w0, int h0, void *ws)
{
    int w = w0, h = h0;
    const size_t n0 = (size_t)w * h;
    float *img[2], *nxt[2], *xyb[2];
    float *cur = (float *)ws;
    for (int i = 0; i < 2; ++i) { img[i] = cur; cur += 3 * n0; nxt[i] = cur; cur += 3 * n0; xyb[i] = cur; cur += 3 * n0; }
    memcpy(img[0], ref_lin, 3 * n0 * 4); memcpy(img[1], dis_lin, 3 * n0 * 4);
    float *mul = cur, *temp = cur + n0;
    float *s11 = cur + 2 * n0, *s22 = cur + 3 * n0, *s12 = cur + 4 * n0, *mu1 = cur + 5 * n0, *mu2 = cur + 6 * n0;
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
    return ssim;
}

double tmo_cpu_path_score_linear(const float *ref_lin, const float *dis_lin, int w0, int h0)
{
    void *ws = malloc(tmo_cpu_path_ws_bytes(w0, h0));
    if (!ws) return NAN;
    const double s = tmo_cpu_path_score_linear_ws(ref_lin, dis_lin, w0, h0, ws);
    free(ws);
    return s;
}

/* ---- frame-level parallel run of the path (SURVEY 8d "CPU baseline beside it": all host cores, one pair per worker) -----------
 * n_pairs frame pairs of ONE decoded 4:2:0 surface format; pair i takes surface pair i % n_inputs.  Every worker thread converts
 * its pair to linear RGB (the oracle's restatement of the reference conversion kernel -- what the GPU path does before the metric)
 * and runs the restated CPU path on it, with buffers allocated ONCE per worker; pairs are handed out by an atomic counter.
 * Returns the wall-clock seconds from the moment all workers stand at the start line to the last one finishing (negative on
 * failure); scores[i] (optional) receives pair i's score. */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>

int tmo_yuv420_biplanar_to_linear(const void *ybase, const void *uvbase, size_t pitch, int w, int h, int bits, int matrix, float *lin);

typedef struct {
    const void *const *ref, *const *dis; /* n_inputs surfaces each: luma rows at `pitch`, CbCr rows from row `coded_h` on */
    int n_inputs;
    size_t pitch;
    int coded_h, w, h, bits, n_pairs;
    double *scores;
    atomic_int next, failed, ready, go;
} tmo_run_t;

static void tmo_nap(void) { const struct timespec ts = {0, 200000}; nanosleep(&ts, NULL); }

static void *tmo_run_worker(void *arg)
{
    tmo_run_t *r = (tmo_run_t *)arg;
    const size_t n = (size_t)r->w * r->h;
    float *lr = malloc(3 * n * sizeof(float)), *ld = malloc(3 * n * sizeof(float));
    void *ws = malloc(tmo_cpu_path_ws_bytes(r->w, r->h));
    if (lr && ld && ws) { /* touch everything before the clock starts: the run measures the path, not first-touch page faults */
        memset(lr, 0, 3 * n * sizeof(float)); memset(ld, 0, 3 * n * sizeof(float)); memset(ws, 0, tmo_cpu_path_ws_bytes(r->w, r->h));
    } else atomic_store(&r->failed, 1);
    atomic_fetch_add(&r->ready, 1);
    while (!atomic_load(&r->go)) tmo_nap(); /* the start line */
    if (!atomic_load(&r->failed))
        for (;;) {
            const int i = atomic_fetch_add(&r->next, 1);
            if (i >= r->n_pairs) break;
            const char *a = (const char *)r->ref[i % r->n_inputs], *b = (const char *)r->dis[i % r->n_inputs];
            tmo_yuv420_biplanar_to_linear(a, a + r->pitch * (size_t)r->coded_h, r->pitch, r->w, r->h, r->bits, 0, lr);
            tmo_yuv420_biplanar_to_linear(b, b + r->pitch * (size_t)r->coded_h, r->pitch, r->w, r->h, r->bits, 0, ld);
            const double s = tmo_cpu_path_score_linear_ws(lr, ld, r->w, r->h, ws);
            if (r->scores) r->scores[i] = s;
        }
    free(lr); free(ld); free(ws);
    return NULL;
}

double tmo_cpu_path_run(const void *const *ref, const void *const *dis, int n_inputs, size_t pitch, int coded_h, int w, int h, int bits,
                        int n_pairs, int n_threads, double *scores)
{
    if (n_inputs < 1 || n_pairs < 1 || n_threads < 1 || n_threads > 4096 || (bits != 8 && bits != 16)) return -1.0;
    tmo_run_t r;
    r.ref = ref; r.dis = dis; r.n_inputs = n_inputs; r.pitch = pitch; r.coded_h = coded_h; r.w = w; r.h = h; r.bits = bits; r.n_pairs = n_pairs; r.scores = scores;
    atomic_init(&r.next, 0); atomic_init(&r.failed, 0); atomic_init(&r.ready, 0); atomic_init(&r.go, 0);
    pthread_t *th = malloc((size_t)n_threads * sizeof *th);
    if (!th) return -1.0;
    int started = 0;
    for (; started < n_threads; ++started)
        if (pthread_create(&th[started], NULL, tmo_run_worker, &r)) break;
    if (started < n_threads) atomic_store(&r.failed, 1); /* not the run that was asked for: the workers that exist leave at once */
    while (atomic_load(&r.ready) < started) tmo_nap();
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    atomic_store(&r.go, 1);
    for (int k = 0; k < started; ++k) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    if (atomic_load(&r.failed)) return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
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
