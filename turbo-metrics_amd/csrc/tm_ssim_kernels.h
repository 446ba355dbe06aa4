// tm_ssim_kernels.h -- gfx950 kernels for SSIM and MS-SSIM of the u8-quantised linear-RGB pair: the inputs of the
// reference's nppiSSIM_8u_C3R_Ctx / nppiWMSSSIM_8u_C3R_Ctx calls (crates/turbo-metrics/src/lib.rs:296-340).
// NPP is closed source and nothing in the reference pins its results, so the arithmetic here is BUILD-DEFINED (the
// published algorithms): 11x11 Gaussian window (sigma 1.5) over the windows that fit inside the
// image, K1 = 0.01, K2 = 0.03, L = 255; five dyadic scales (2x2 box mean) for MS-SSIM.
//
//   k_ssim_down<SRC_U8>  grid (ceil(dw/64), dh, slots*2*3)  block 64     one level of the box pyramid
//   k_ssim_stats<SRC_U8> grid (tiles_x, tiles_y, slots*3)   block 256    32x32 windows per workgroup: 42x42 input tile of both
//                                                                      sides in LDS, row filter of {x, y, x^2, y^2, xy} into
//                                                                      LDS, column filter + SSIM / cs terms from LDS, f64 sums
//   k_ssim_finish        grid (slots)                       block 32     fixed-order sum of the tile partials -> 30 sums / slot
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
    float g[TM_SSIM_TAPS];
};

static inline void tm_make_ssim_geom(TmSsimGeom *s, int w, int h, const float g[TM_SSIM_TAPS])
{
    unsigned long long off = 0;
    s->tile_off[0] = 0;
    for (int i = 0; i < TM_SSIM_SCALES; ++i) {
        s->w[i] = w; s->h[i] = h;
        s->pitch[i] = tm_round_up(w > 0 ? w : 1, 64);
        s->off[i] = off;
        if (i > 0) off += (unsigned long long)(h > 0 ? h : 1) * s->pitch[i];
        const int ow = w - 10, oh = h - 10;
        s->tiles_x[i] = ow > 0 ? (ow + 31) / 32 : 0;
        s->tiles_y[i] = oh > 0 ? (oh + 31) / 32 : 0;
        s->tile_off[i + 1] = s->tile_off[i] + s->tiles_x[i] * s->tiles_y[i];
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

// 2x2 box mean with decimation; an odd last row / column is dropped: ((a + b) + (c + d)) * 0.25
template <bool SRC_U8>
__global__ void __launch_bounds__(64) k_ssim_down(TmSsimGeom sg, int s, const unsigned char *__restrict__ Q, float *__restrict__ PYR)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= sg.w[s]) return;
    const int img = blockIdx.z / 3, c = blockIdx.z % 3;
    float a, b, cc, d;
    if (SRC_U8) {
        const unsigned char *p = Q + ((size_t)img * 3 + c) * sg.qplane + (size_t)(2 * y) * sg.pitch[0] + 2 * x;
        a = (float)p[0]; b = (float)p[1]; cc = (float)p[sg.pitch[0]]; d = (float)p[sg.pitch[0] + 1];
    } else {
        const float *p = ssim_plane_f(sg, PYR, img, c, s - 1) + (size_t)(2 * y) * sg.pitch[s - 1] + 2 * x;
        a = p[0]; b = p[1]; cc = p[sg.pitch[s - 1]]; d = p[sg.pitch[s - 1] + 1];
    }
    float *o = const_cast<float *>(ssim_plane_f(sg, PYR, img, c, s)) + (size_t)y * sg.pitch[s] + x;
    *o = ((a + b) + (cc + d)) * 0.25f;
}

template <bool SRC_U8>
__global__ void __launch_bounds__(256) k_ssim_stats(TmSsimGeom sg, int s, const unsigned char *__restrict__ Q,
                                                    const float *__restrict__ PYR, double *__restrict__ PART)
{
    __shared__ float in[2][42][43];
    __shared__ float hz[5][42][33];
    __shared__ double red[2][4];
    const int tid = threadIdx.x;
    const int slot = blockIdx.z / 3, c = blockIdx.z % 3;
    const int w = sg.w[s], h = sg.h[s];
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    // ---- 42 x 42 input tile of both sides (samples outside the image read as 0; no valid window uses them)
    for (int i = tid; i < 2 * 42 * 42; i += 256) {
        const int side = i / (42 * 42), r = (i % (42 * 42)) / 42, col = i % 42;
        const int x = x0 + col, y = y0 + r;
        float v = 0.0f;
        if (x < w && y < h) {
            if (SRC_U8) v = (float)Q[((size_t)(slot * 2 + side) * 3 + c) * sg.qplane + (size_t)y * sg.pitch[0] + x];
            else v = ssim_plane_f(sg, PYR, slot * 2 + side, c, s)[(size_t)y * sg.pitch[s] + x];
        }
        in[side][r][col] = v;
    }
    __syncthreads();
    // ---- row filter of x, y, x^2, y^2, xy: 42 rows x 32 columns
    for (int i = tid; i < 42 * 32; i += 256) {
        const int r = i >> 5, col = i & 31;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f;
#pragma unroll
        for (int k = 0; k < TM_SSIM_TAPS; ++k) {
            const float rv = in[0][r][col + k], dv = in[1][r][col + k], gk = sg.g[k];
            a0 = __builtin_fmaf(gk, rv, a0);
            a1 = __builtin_fmaf(gk, dv, a1);
            a2 = __builtin_fmaf(gk, rv * rv, a2);
            a3 = __builtin_fmaf(gk, dv * dv, a3);
            a4 = __builtin_fmaf(gk, rv * dv, a4);
        }
        hz[0][r][col] = a0; hz[1][r][col] = a1; hz[2][r][col] = a2; hz[3][r][col] = a3; hz[4][r][col] = a4;
    }
    __syncthreads();
    // ---- column filter + the two terms, 4 windows per lane
    const float C1 = 6.5025f, C2 = 58.5225f; // (0.01*255)^2, (0.03*255)^2
    double s_ssim = 0.0, s_cs = 0.0;
    const int col = tid & 31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (tid >> 5) + 8 * j;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f;
#pragma unroll
        for (int k = 0; k < TM_SSIM_TAPS; ++k) {
            const float gk = sg.g[k];
            a0 = __builtin_fmaf(gk, hz[0][r + k][col], a0);
            a1 = __builtin_fmaf(gk, hz[1][r + k][col], a1);
            a2 = __builtin_fmaf(gk, hz[2][r + k][col], a2);
            a3 = __builtin_fmaf(gk, hz[3][r + k][col], a3);
            a4 = __builtin_fmaf(gk, hz[4][r + k][col], a4);
        }
        const float mxx = a0 * a0, myy = a1 * a1, mxy = a0 * a1;
        const float sx = a2 - mxx, sy = a3 - myy, sxy = a4 - mxy;
        const float cs = __builtin_fmaf(2.0f, sxy, C2) / ((sx + sy) + C2);
        const float l = __builtin_fmaf(2.0f, mxy, C1) / ((mxx + myy) + C1);
        if (x0 + col < w - 10 && y0 + r < h - 10) {
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
            double *o = PART + (((size_t)slot * 3 + c) * sg.tile_off[TM_SSIM_SCALES] + sg.tile_off[s] + blockIdx.y * sg.tiles_x[s] + blockIdx.x) * 2;
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
        double *o = PART + (((size_t)slot * 3 + c) * sg.tile_off[TM_SSIM_SCALES] + sg.tile_off[s] + blockIdx.y * sg.tiles_x[s] + blockIdx.x) * 2;
        o[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        o[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
#endif
}

// SUMS[slot][channel 3][scale 5][ssim, cs]: tiles added in index order (deterministic run to run)
__global__ void __launch_bounds__(32) k_ssim_finish(TmSsimGeom sg, const double *__restrict__ PART, double *__restrict__ SUMS)
{
    const int i = threadIdx.x, slot = blockIdx.x;
    if (i >= 30) return;
    const int c = i / 10, s = (i % 10) / 2, which = i & 1;
    double sum = 0.0;
    for (int t = sg.tile_off[s]; t < sg.tile_off[s + 1]; ++t)
        sum += PART[(((size_t)slot * 3 + c) * sg.tile_off[TM_SSIM_SCALES] + t) * 2 + which];
    SUMS[(size_t)slot * 30 + i] = sum;
}

} // namespace tmk
