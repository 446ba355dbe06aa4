// tm_ssim_kernels.h -- gfx950 kernels for SSIM and MS-SSIM of the u8-quantised linear-RGB pair: the inputs of the
// reference's nppiSSIM_8u_C3R_Ctx / nppiWMSSSIM_8u_C3R_Ctx calls (crates/turbo-metrics/src/lib.rs:296-340).
// NPP is closed source and nothing in the reference pins its results, so the arithmetic here is BUILD-DEFINED (the
// published algorithms): 11x11 Gaussian window (sigma 1.5) over the windows that fit inside the
// image, K1 = 0.01, K2 = 0.03, L = 255; five dyadic scales (2x2 box mean) for MS-SSIM.
//
//   k_ssim_pyramid  grid (ceil(w/32), ceil(h/32), slots*2*3)  block 256   scales 1..4 of the box pyramid from one 32x32 u8 tile
//   k_ssim_stats    grid (slots*3, tiles of all scales)       block 256   32x32 windows per workgroup: 42x42 input tile of both
//                                                                       sides in LDS, row filter of {x, y, x^2, y^2, xy} into
//                                                                       LDS, column filter + SSIM / cs terms from LDS, f64 sums
//   k_ssim_finish   grid (slots, 30)                          block 64    fixed-order sum of the tile partials -> 30 sums / slot
// Filter order (the oracle executes the same): taps ascending, acc = fma(g[k], v, acc) starting from 0.
#pragma once
#include "tm_device_math.h"
#include "tm_geom.h"

#define TM_SSIM_SCALES 5
#define TM_SSIM_TAPS 11

struct TmSsimGeom {
    int w[TM_SSIM_SCALES], h[TM_SSIM_SCALES];
    int pitch[TM_SSIM_SCALES];               // elements (bytes at scale 0, floats above)
    unsigned long long off[TM_SSIM_SCALES];  // float offset of scales 1..4 inside one (slot, side, channel) pyramid; [0] unused
    unsigned long long qplane;               // bytes of one u8 plane (scale 0)
    unsigned long long pyr;                  // floats of scales 1..4 of one (slot, side, channel)
    int tiles_x[TM_SSIM_SCALES], tiles_y[TM_SSIM_SCALES];
    int tile_off[TM_SSIM_SCALES + 1];        // prefix sums of tiles per (slot, channel)
    // streaming kernel: one wave = one strip of TM_SSIM_STRIP window columns x one segment of TM_SSIM_SEG window rows
    int strips_x[TM_SSIM_SCALES], segs_y[TM_SSIM_SCALES];
    int item_off[TM_SSIM_SCALES + 1];        // prefix sums of (strip, segment) items per (slot, channel)
    float g[TM_SSIM_TAPS];
};
#define TM_SSIM_STRIP 54  /* 64 lanes hold 64 input columns = 54 windows + 10 columns of halo */
#define TM_SSIM_SEG 128   /* window rows per wave (each wave re-reads 10 halo rows) */

static inline void tm_make_ssim_geom(TmSsimGeom *s, int w, int h, const float g[TM_SSIM_TAPS])
{
    unsigned long long off = 0;
    s->tile_off[0] = 0;
    s->item_off[0] = 0;
    for (int i = 0; i < TM_SSIM_SCALES; ++i) {
        s->w[i] = w; s->h[i] = h;
        s->pitch[i] = tm_round_up(w > 0 ? w : 1, 64);
        s->off[i] = off;
        if (i > 0) off += (unsigned long long)(h > 0 ? h : 1) * s->pitch[i];
        const int ow = w - 10, oh = h - 10;
        s->tiles_x[i] = ow > 0 ? (ow + 31) / 32 : 0;
        s->tiles_y[i] = oh > 0 ? (oh + 31) / 32 : 0;
        s->tile_off[i + 1] = s->tile_off[i] + s->tiles_x[i] * s->tiles_y[i];
        s->strips_x[i] = ow > 0 ? (ow + TM_SSIM_STRIP - 1) / TM_SSIM_STRIP : 0;
        s->segs_y[i] = oh > 0 ? (oh + TM_SSIM_SEG - 1) / TM_SSIM_SEG : 0;
        s->item_off[i + 1] = s->item_off[i] + s->strips_x[i] * s->segs_y[i];
        w /= 2; h /= 2;
    }
    s->qplane = (unsigned long long)s->h[0] * s->pitch[0];
    s->pyr = off;
    for (int k = 0; k < TM_SSIM_TAPS; ++k) s->g[k] = g[k];
}

namespace tmk {

// plane of scale `s` of image (slot*2+side), channel c
__device__ __forceinline__ const float *ssim_plane_f(const TmSsimGeom &sg, const float *PYR, int img, int c, int s)
{
    return PYR + ((size_t)img * 3 + c) * sg.pyr + sg.off[s];
}

// 2x2 box mean with decimation, four levels at once; an odd last row / column of a level is dropped (level s has
// floor(w/2^s) x floor(h/2^s) pixels): ((a + b) + (c + d)) * 0.25.  Tiles are 32-aligned, so every parent stays inside.
__global__ void __launch_bounds__(256) k_ssim_pyramid(TmSsimGeom sg, const unsigned char *__restrict__ Q, float *__restrict__ PYR)
{
    __shared__ float l1[16][17], l2[8][9], l3[4][5];
    const int tid = threadIdx.x;
    const int img = blockIdx.z / 3, c = blockIdx.z % 3;
    const int bx = blockIdx.x, by = blockIdx.y;
    {
        const int tx = tid & 15, ty = tid >> 4;
        const int x = bx * 16 + tx, y = by * 16 + ty;
        float v = 0.0f;
        if (x < sg.w[1] && y < sg.h[1]) {
            const unsigned char *p = Q + ((size_t)img * 3 + c) * sg.qplane + (size_t)(2 * y) * sg.pitch[0] + 2 * x;
            const float a = (float)p[0], b = (float)p[1], cc = (float)p[sg.pitch[0]], d = (float)p[sg.pitch[0] + 1];
            v = ((a + b) + (cc + d)) * 0.25f;
            const_cast<float *>(ssim_plane_f(sg, PYR, img, c, 1))[(size_t)y * sg.pitch[1] + x] = v;
        }
        l1[ty][tx] = v;
    }
    __syncthreads();
    if (tid < 64) {
        const int tx = tid & 7, ty = tid >> 3;
        const int x = bx * 8 + tx, y = by * 8 + ty;
        const float v = ((l1[2 * ty][2 * tx] + l1[2 * ty][2 * tx + 1]) + (l1[2 * ty + 1][2 * tx] + l1[2 * ty + 1][2 * tx + 1])) * 0.25f;
        l2[ty][tx] = v;
        if (x < sg.w[2] && y < sg.h[2]) const_cast<float *>(ssim_plane_f(sg, PYR, img, c, 2))[(size_t)y * sg.pitch[2] + x] = v;
    }
    __syncthreads();
    if (tid < 16) {
        const int tx = tid & 3, ty = tid >> 2;
        const int x = bx * 4 + tx, y = by * 4 + ty;
        const float v = ((l2[2 * ty][2 * tx] + l2[2 * ty][2 * tx + 1]) + (l2[2 * ty + 1][2 * tx] + l2[2 * ty + 1][2 * tx + 1])) * 0.25f;
        l3[ty][tx] = v;
        if (x < sg.w[3] && y < sg.h[3]) const_cast<float *>(ssim_plane_f(sg, PYR, img, c, 3))[(size_t)y * sg.pitch[3] + x] = v;
    }
    __syncthreads();
    if (tid < 4) {
        const int tx = tid & 1, ty = tid >> 1;
        const int x = bx * 2 + tx, y = by * 2 + ty;
        const float v = ((l3[2 * ty][2 * tx] + l3[2 * ty][2 * tx + 1]) + (l3[2 * ty + 1][2 * tx] + l3[2 * ty + 1][2 * tx + 1])) * 0.25f;
        if (x < sg.w[4] && y < sg.h[4]) const_cast<float *>(ssim_plane_f(sg, PYR, img, c, 4))[(size_t)y * sg.pitch[4] + x] = v;
    }
}

__global__ void __launch_bounds__(256) k_ssim_stats(TmSsimGeom sg, int nscales, const unsigned char *__restrict__ Q,
                                                    const float *__restrict__ PYR, double *__restrict__ PART)
{
    __shared__ float in[2][42][44];
    __shared__ float hz[5][42][33];
    __shared__ double red[2][4];
    const int tid = threadIdx.x;
    const int slot = blockIdx.x / 3, c = blockIdx.x % 3;
    int s = 0; // scale of this tile: all scales share the launch (grid.y runs over sg.tile_off[nscales] tiles)
#pragma unroll
    for (int i = 1; i < TM_SSIM_SCALES; ++i)
        if (i < nscales && (int)blockIdx.y >= sg.tile_off[i]) s = i;
    const int tile = (int)blockIdx.y - sg.tile_off[s];
    const int w = sg.w[s], h = sg.h[s];
    const int x0 = (tile % sg.tiles_x[s]) * 32, y0 = (tile / sg.tiles_x[s]) * 32;
    // ---- 42 x 42 input tile of both sides, one row per wave-load (lanes = columns); samples outside the image read
    // as 0 (no valid window uses them)
    {
        // all 21 row loads of a wave are issued before the first LDS write (a load followed by its own write in a loop
        // waits out the full memory latency 21 times)
        const int lane = tid & 63, wave = tid >> 6;
        const int x = x0 + lane;
        float v[21];
#pragma unroll
        for (int k = 0; k < 21; ++k) {
            const int i = wave + 4 * k, side = i / 42, r = i % 42;
            const int y = y0 + r;
            v[k] = 0.0f;
            if (lane < 42 && x < w && y < h) {
                if (s == 0) v[k] = (float)Q[((size_t)(slot * 2 + side) * 3 + c) * sg.qplane + (size_t)y * sg.pitch[0] + x];
                else v[k] = ssim_plane_f(sg, PYR, slot * 2 + side, c, s)[(size_t)y * sg.pitch[s] + x];
            }
        }
        if (lane < 42) {
#pragma unroll
            for (int k = 0; k < 21; ++k) {
                const int i = wave + 4 * k;
                in[i / 42][i % 42][lane] = v[k];
            }
        }
    }
    if (tid < 2 * 42) { in[tid / 42][tid % 42][42] = 0.0f; in[tid / 42][tid % 42][43] = 0.0f; }
    __syncthreads();
    // ---- row filter of x, y, x^2, y^2, xy: 42 rows x 32 columns, four neighbouring columns per lane (14 samples of each
    // side serve 4 x 11 taps: 7 LDS reads per output instead of 22)
    for (int i = tid; i < 42 * 8; i += 256) {
        const int r = i >> 3, c0 = (i & 7) * 4;
        float rv[14], dv[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) { rv[k] = in[0][r][c0 + k]; dv[k] = in[1][r][c0 + k]; }
        float rr[14], dd[14], rd[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) { rr[k] = rv[k] * rv[k]; dd[k] = dv[k] * dv[k]; rd[k] = rv[k] * dv[k]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f;
#pragma unroll
            for (int k = 0; k < TM_SSIM_TAPS; ++k) {
                const float gk = sg.g[k];
                a0 = __builtin_fmaf(gk, rv[j + k], a0);
                a1 = __builtin_fmaf(gk, dv[j + k], a1);
                a2 = __builtin_fmaf(gk, rr[j + k], a2);
                a3 = __builtin_fmaf(gk, dd[j + k], a3);
                a4 = __builtin_fmaf(gk, rd[j + k], a4);
            }
            hz[0][r][c0 + j] = a0; hz[1][r][c0 + j] = a1; hz[2][r][c0 + j] = a2; hz[3][r][c0 + j] = a3; hz[4][r][c0 + j] = a4;
        }
    }
    __syncthreads();
    // ---- column filter + the two terms: four vertically neighbouring windows per lane (14 rows serve 4 x 11 taps)
    const float C1 = 6.5025f, C2 = 58.5225f; // (0.01*255)^2, (0.03*255)^2
    double s_ssim = 0.0, s_cs = 0.0;
    const int col = tid & 31, r0 = (tid >> 5) * 4;
    float acc[4][5];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[j][q] = 0.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) v[k] = hz[q][r0 + k][col];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < TM_SSIM_TAPS; ++k) acc[j][q] = __builtin_fmaf(sg.g[k], v[j + k], acc[j][q]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = acc[j][0], a1 = acc[j][1], a2 = acc[j][2], a3 = acc[j][3], a4 = acc[j][4];
        const float mxx = a0 * a0, myy = a1 * a1, mxy = a0 * a1;
        const float sx = a2 - mxx, sy = a3 - myy, sxy = a4 - mxy;
        const float cs = __builtin_fmaf(2.0f, sxy, C2) / ((sx + sy) + C2);
        const float l = __builtin_fmaf(2.0f, mxy, C1) / ((mxx + myy) + C1);
        if (x0 + col < w - 10 && y0 + r0 + j < h - 10) {
            s_ssim += (double)(l * cs);
            s_cs += (double)cs;
        }
    }
    // ---- workgroup sum: 64-lane shuffle tree per wave, then the four wave totals in a fixed order
#ifdef TM_EMULATE
    {
        __shared__ double all[2][256]; // CPU lane emulation (tests/emul): no shuffles, plain in-order sum
        all[0][tid] = s_ssim; all[1][tid] = s_cs;
        __syncthreads();
        if (tid == 0) {
            double t0 = 0.0, t1 = 0.0;
            for (int i = 0; i < 256; ++i) { t0 += all[0][i]; t1 += all[1][i]; }
            double *o = PART + (((size_t)slot * 3 + c) * sg.tile_off[TM_SSIM_SCALES] + blockIdx.y) * 2;
            o[0] = t0; o[1] = t1;
        }
        (void)red;
    }
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_ssim += __shfl_down(s_ssim, off, 64);
        s_cs += __shfl_down(s_cs, off, 64);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s_ssim; red[1][tid >> 6] = s_cs; }
    __syncthreads();
    if (tid == 0) {
        double *o = PART + (((size_t)slot * 3 + c) * sg.tile_off[TM_SSIM_SCALES] + blockIdx.y) * 2;
        o[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        o[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Streaming statistics kernel: one wave walks one strip of 54 window columns down one segment of window rows.
// Lane L owns input column x_base + L.  Per input row: both samples go through a 2 x 76-float LDS row so that every lane
// can read its 11 right-hand neighbours (wave-synchronous: no workgroup barrier anywhere), the row filter of
// {x, y, x^2, y^2, xy} is evaluated for the lane's column and pushed into an 11-row register window; once the window is
// full every step also evaluates the column filter over it and the two terms of one window.  Same operations in the same
// order as k_ssim_stats (taps ascending, fma from 0) -> the same per-window values; only the order of the f64 sums differs.
// grid (slots*3, items of all scales), block 64.  PART[(slot*3+c)][item][2].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_ssim_stream(TmSsimGeom sg, int nscales, const unsigned char *__restrict__ Q,
                                                    const float *__restrict__ PYR, double *__restrict__ PART)
{
    __shared__ float row[2][80];
    const int lane = threadIdx.x;
    const int slot = blockIdx.x / 3, c = blockIdx.x % 3;
    int s = 0;
#pragma unroll
    for (int i = 1; i < TM_SSIM_SCALES; ++i)
        if (i < nscales && (int)blockIdx.y >= sg.item_off[i]) s = i;
    const int item = (int)blockIdx.y - sg.item_off[s];
    const int w = sg.w[s], h = sg.h[s];
    const int x_base = (item % sg.strips_x[s]) * TM_SSIM_STRIP, y_base = (item / sg.strips_x[s]) * TM_SSIM_SEG;
    const int oh = h - 10;
    const int y_end = min(y_base + TM_SSIM_SEG, oh); // window rows [y_base, y_end)
    const int x = x_base + lane;
    const bool in_x = x < w;
    const bool out_x = lane < TM_SSIM_STRIP && x < w - 10;
    const unsigned char *qr = Q + ((size_t)(slot * 2 + 0) * 3 + c) * sg.qplane + (in_x ? x : 0);
    const unsigned char *qd = Q + ((size_t)(slot * 2 + 1) * 3 + c) * sg.qplane + (in_x ? x : 0);
    const float *fr = s ? ssim_plane_f(sg, PYR, slot * 2 + 0, c, s) + (in_x ? x : 0) : nullptr;
    const float *fd = s ? ssim_plane_f(sg, PYR, slot * 2 + 1, c, s) + (in_x ? x : 0) : nullptr;
    const int pitch = sg.pitch[s];
    auto load = [&](int y, float &a, float &b) { // row y of both sides at this lane's column (rows past the image: 0)
        const int yc = y < h ? y : h - 1;
        float va, vb;
        if (s == 0) { va = (float)qr[(size_t)yc * pitch]; vb = (float)qd[(size_t)yc * pitch]; }
        else { va = fr[(size_t)yc * pitch]; vb = fd[(size_t)yc * pitch]; }
        const bool ok = in_x && y < h;
        a = ok ? va : 0.0f; b = ok ? vb : 0.0f;
    };
    constexpr int PF = 11; // rows of load prefetch (= the window depth, so that one unroll of 11 makes every slot static)
    float pa[PF], pb[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) load(y_base + k, pa[k], pb[k]);
    if (lane < 16) { row[0][64 + lane] = 0.0f; row[1][64 + lane] = 0.0f; } // the halo lanes' right-hand neighbours
    float win[11][5];
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) win[k][q] = 0.0f;
    const float C1 = 6.5025f, C2 = 58.5225f; // (0.01*255)^2, (0.03*255)^2
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int n_rows = (y_end - y_base) + 10; // input rows y_base .. y_end+9
    float g[TM_SSIM_TAPS];
#pragma unroll
    for (int k = 0; k < TM_SSIM_TAPS; ++k) g[k] = sg.g[k];
    for (int t0 = 0; t0 < n_rows; t0 += 11) {
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int t = t0 + j;
            if (t < n_rows) { // wave-uniform
                const float rv = pa[j % PF], dv = pb[j % PF];
                load(y_base + t + PF, pa[j % PF], pb[j % PF]);
                __builtin_amdgcn_wave_barrier();
                row[0][lane] = rv; row[1][lane] = dv;
                __builtin_amdgcn_wave_barrier();
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f;
#pragma unroll
                for (int k = 0; k < TM_SSIM_TAPS; ++k) {
                    const float r = row[0][lane + k], d = row[1][lane + k], gk = g[k];
                    a0 = __builtin_fmaf(gk, r, a0);
                    a1 = __builtin_fmaf(gk, d, a1);
                    a2 = __builtin_fmaf(gk, r * r, a2);
                    a3 = __builtin_fmaf(gk, d * d, a3);
                    a4 = __builtin_fmaf(gk, r * d, a4);
                }
                win[j % 11][0] = a0; win[j % 11][1] = a1; win[j % 11][2] = a2; win[j % 11][3] = a3; win[j % 11][4] = a4;
                if (t >= 10) { // window rows t-10 .. t are in slots (j+1)%11 .. j%11
                    float v[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int k = 0; k < TM_SSIM_TAPS; ++k)
#pragma unroll
                        for (int q = 0; q < 5; ++q) v[q] = __builtin_fmaf(g[k], win[(j + 1 + k) % 11][q], v[q]);
                    const float mxx = v[0] * v[0], myy = v[1] * v[1], mxy = v[0] * v[1];
                    const float sx = v[2] - mxx, sy = v[3] - myy, sxy = v[4] - mxy;
                    const float cs = __builtin_fmaf(2.0f, sxy, C2) / ((sx + sy) + C2);
                    const float l = __builtin_fmaf(2.0f, mxy, C1) / ((mxx + myy) + C1);
                    if (out_x) { acc[0] += (double)(l * cs); acc[1] += (double)cs; }
                }
            }
        }
    }
#ifdef TM_EMULATE
    { // CPU lane emulation runs the 64 lanes as concurrent host threads: sum through memory, not through shuffles
        __shared__ double redl[2][64];
        redl[0][lane] = acc[0]; redl[1][lane] = acc[1];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double t0 = 0.0, t1 = 0.0;
            for (int i = 0; i < 64; ++i) { t0 += redl[0][i]; t1 += redl[1][i]; }
            double *o = PART + (((size_t)slot * 3 + c) * sg.item_off[TM_SSIM_SCALES] + blockIdx.y) * 2;
            o[0] = t0; o[1] = t1;
        }
        __builtin_amdgcn_wave_barrier();
    }
#else
    if (tm_wave_sum6(acc)) {
        double *o = PART + (((size_t)slot * 3 + c) * sg.item_off[TM_SSIM_SCALES] + blockIdx.y) * 2;
        o[0] = acc[0]; o[1] = acc[1];
    }
#endif
}

// SUMS[slot][channel 3][scale 5][ssim, cs]: lane l adds tiles l, l+64, ... in order, then the 64 lane totals are added
// by a fixed tree (deterministic run to run).  grid (slots, 30), block 64.
__global__ void __launch_bounds__(64) k_ssim_finish(TmSsimGeom sg, int streamed, const double *__restrict__ PART, double *__restrict__ SUMS)
{
    const int slot = blockIdx.x, i = blockIdx.y;
    const int c = i / 10, s = (i % 10) / 2, which = i & 1;
    const int *off = streamed ? sg.item_off : sg.tile_off; // partials of k_ssim_stream or of k_ssim_stats
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int t = off[s] + (int)threadIdx.x; t < off[s + 1]; t += 64)
        a[0] += PART[(((size_t)slot * 3 + c) * off[TM_SSIM_SCALES] + t) * 2 + which];
    if (tm_wave_sum6(a)) SUMS[(size_t)slot * 30 + i] = a[0];
}

} // namespace tmk
