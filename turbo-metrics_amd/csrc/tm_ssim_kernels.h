// tm_ssim_kernels.h -- gfx950 kernels for SSIM and MS-SSIM of the u8-quantised linear-RGB pair: the inputs of the
// reference's nppiSSIM_8u_C3R_Ctx / nppiWMSSSIM_8u_C3R_Ctx calls (crates/turbo-metrics/src/lib.rs:296-340).
// NPP is closed source and nothing in the reference pins its results, so the arithmetic here is BUILD-DEFINED (the
// published algorithms; stated operation by operation in oracle/tm_ssim.c): 11x11 Gaussian window
// (sigma 1.5) over the windows that fit inside the image, K1 = 0.01, K2 = 0.03, L = 255; five dyadic scales (2x2 box mean,
// odd last row / column dropped) for MS-SSIM.
//
//   k_ssim_pyramid  grid (ceil(w/128), ceil(h/32), slots*2*3) block 64    scales 1..4 of the box-sum pyramid, lane = 8x8 pixels
//   k_ssim_stream   grid (slots*3, items of all scales)       block 64    one wave = a strip of 118 window columns x a segment
//                                                                         of window rows, see below
//   k_ssim_finish   grid (slots, 30)                          block 64    fixed-order sum of the partials -> 30 sums / slot
//
// The pyramid is stored as INTEGER box sums (u16): a pixel of scale s is ((a + b) + (c + d)) * 0.25 of scale s-1 in the oracle,
// and since every value is a multiple of 4^-s below 256 all of those f32 operations are exact -- the pixel equals
// (sum of its 4^s u8 samples) * 4^-s, and the sum fits 16 bits (<= 255 * 256).  Half the bytes of an f32 pyramid, same bits.
//
// This stage is bound by arithmetic, not by HBM.  SSIM needs four window means per channel -- E[x], E[y], E[xy] and
// E[x^2 + y^2]: the variances only ever appear as their sum, sigma_x^2 + sigma_y^2 = E[x^2 + y^2] - mu_x^2 - mu_y^2 -- so the
// separable 11-tap filter costs 88 fused multiply-adds per window and channel, against 2 bytes of input.  The kernel is built
// around the VALU:
//   * lane = TWO adjacent image columns, walking down its segment one input row per step: the per-sample quantities
//     s = fma(x, x, y * y) and p = x * y are formed ONCE per sample by the lane that loaded it and travel to the five
//     neighbour lanes that need them with the sample itself: {x, y, s, p} is one 16-byte LDS element per column;
//   * everything is said on register pairs so that the filters run as v_pk_fma_f32 (two fused multiply-adds per
//     instruction): row filter and column filter on the {x, y} and {s, p} pairs of one column, the SSIM terms on
//     {col 0, col 1} pairs;
//   * neighbours' columns come from a 2.3-KB wave-private LDS row as ten conflict-free ds_read_b128 per row (even and odd
//     columns in separate arrays: lane stride 16 bytes; no workgroup barrier: LDS operations of one wave execute in order);
//     the row-filtered values live in an 11-row register window (static slots through an unroll of 11), the column filter
//     reads registers only;
//   * need_l: the luminance term l (one of the two IEEE divisions per window) is only evaluated where its sum is used --
//     MS-SSIM uses the contrast-structure term alone on scales 0..3 (Wang et al. 2003); tm_engine_set_full_sums(e, 1)
//     evaluates everything.
// Filter order (the oracle executes the same): rows first, taps ascending, acc = fma(g[k], v, acc) starting from 0.
#pragma once
#include "tm_device_math.h"
#include "tm_geom.h"

#define TM_SSIM_SCALES 5
#define TM_SSIM_TAPS 11
#define TM_SSIM_STRIP 118  /* window columns per wave: 64 lanes x 2 columns = 128 input columns, 10 of them halo */
#define TM_SSIM_SEG 192    /* upper bound of the window rows per wave (each wave re-reads 10 halo rows); segments are balanced */

struct TmSsimGeom {
    int w[TM_SSIM_SCALES], h[TM_SSIM_SCALES];
    int pitch[TM_SSIM_SCALES];               // elements per row: bytes at scale 0 (u8), u16 sums above
    unsigned long long off[TM_SSIM_SCALES];  // u16 offset of scales 1..4 inside one (slot, side, channel) pyramid; [0] unused
    unsigned long long qplane;               // bytes of one u8 plane (scale 0)
    unsigned long long pyr;                  // u16 elements of scales 1..4 of one (slot, side, channel)
    int strips_x[TM_SSIM_SCALES], segs_y[TM_SSIM_SCALES], seg_rows[TM_SSIM_SCALES];
    int item_off[TM_SSIM_SCALES + 1];        // prefix sums of (strip, segment) items per (slot, channel)
    float g[TM_SSIM_TAPS];
};

static inline void tm_make_ssim_geom(TmSsimGeom *s, int w, int h, const float g[TM_SSIM_TAPS])
{
    unsigned long long off = 0;
    s->item_off[0] = 0;
    for (int i = 0; i < TM_SSIM_SCALES; ++i) {
        s->w[i] = w; s->h[i] = h;
        s->pitch[i] = tm_round_up(w > 0 ? w : 1, 64);
        s->off[i] = off;
        if (i > 0) off += (unsigned long long)(h > 0 ? h : 1) * s->pitch[i];
        const int ow = w - 10, oh = h - 10;
        s->strips_x[i] = ow > 0 ? (ow + TM_SSIM_STRIP - 1) / TM_SSIM_STRIP : 0;
        s->segs_y[i] = oh > 0 ? (oh + TM_SSIM_SEG - 1) / TM_SSIM_SEG : 0;
        s->seg_rows[i] = s->segs_y[i] > 0 ? (oh + s->segs_y[i] - 1) / s->segs_y[i] : 0;
        s->item_off[i + 1] = s->item_off[i] + s->strips_x[i] * s->segs_y[i];
        w /= 2; h /= 2;
    }
    s->qplane = (unsigned long long)s->h[0] * s->pitch[0];
    s->pyr = off;
    for (int k = 0; k < TM_SSIM_TAPS; ++k) s->g[k] = g[k];
}

namespace tmk {

// plane of scale `s` (1..4) of image (slot*2+side), channel c
__device__ __forceinline__ const unsigned short *ssim_plane_s(const TmSsimGeom &sg, const unsigned short *PYR, int img, int c, int s)
{
    return PYR + ((size_t)img * 3 + c) * sg.pyr + sg.off[s];
}

// Box sums of scales 1..4 in one pass over the u8 plane; an odd last row / column of a level is dropped (level s has
// floor(w/2^s) x floor(h/2^s) pixels).  Lane = one 8 x 8 pixel block, read as eight 8-byte rows: scales 1..3 of the block (4 x 4,
// 2 x 2, 1 sums) are formed in registers with packed 16-bit adds -- two neighbouring sums in one dword, which is also how they
// are stored --, scale 4 from the 2 x 2 lane group through three shuffles.  The wave covers 128 x 32 pixels (lanes 16 x 4), so a
// row of the tile is one whole 128-byte line (read) and the scale-1 sums of a row are one 128-byte line too (written).
// No LDS, no barrier.  grid (ceil(w/128), ceil(h/32), slots*2*3), block 64.
__global__ void __launch_bounds__(64) k_ssim_pyramid(TmSsimGeom sg, const unsigned char *__restrict__ Q, unsigned short *__restrict__ PYR)
{
    const int lane = threadIdx.x;
    const int img = blockIdx.z / 3, c = blockIdx.z % 3;
    const int x0 = blockIdx.x * 128 + 8 * (lane & 15), y0 = blockIdx.y * 32 + 8 * (lane >> 4);
    const bool live = x0 < sg.pitch[0]; // the pitch is a multiple of 64: a live lane's 8 bytes stay inside the (padded) row
    const unsigned char *q = Q + ((size_t)img * 3 + c) * sg.qplane + x0;
    unsigned short *base = PYR + ((size_t)img * 3 + c) * sg.pyr;
    const int h = sg.h[0];
    // h0[r], h1[r]: the four horizontal pair sums of row r, two per dword (16 bits each)
    unsigned l1[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hs[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = min(y0 + 2 * j + i, h - 1); // rows past the image only feed sums that are never stored
            const uint2 u = live ? *(const uint2 *)(q + (size_t)y * sg.pitch[0]) : make_uint2(0u, 0u);
            hs[i][0] = (u.x & 0x00FF00FFu) + ((u.x >> 8) & 0x00FF00FFu);
            hs[i][1] = (u.y & 0x00FF00FFu) + ((u.y >> 8) & 0x00FF00FFu);
        }
        l1[j][0] = hs[0][0] + hs[1][0]; // scale-1 sums of columns x0/2 + {0, 1} | {2, 3}: <= 1020 each, no carry between the halves
        l1[j][1] = hs[0][1] + hs[1][1];
    }
    const int x1 = x0 >> 1, y1 = y0 >> 1, x2 = x0 >> 2, y2 = y0 >> 2, x3 = x0 >> 3, y3 = y0 >> 3;
    if (x1 < sg.w[1]) // x1 is a multiple of 4 and the pitch one of 64: the four sums stay inside the row (columns past w[1] are padding)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (y1 + j < sg.h[1]) *(uint2 *)(base + sg.off[1] + (size_t)(y1 + j) * sg.pitch[1] + x1) = make_uint2(l1[j][0], l1[j][1]);
    unsigned l2[2]; // rows of scale 2: two sums per dword
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned a = l1[2 * j][0] + l1[2 * j + 1][0], b = l1[2 * j][1] + l1[2 * j + 1][1]; // vertical pairs (<= 2040 per half)
        l2[j] = ((a & 0xFFFFu) + (a >> 16)) | (((b & 0xFFFFu) + (b >> 16)) << 16);
    }
    if (x2 < sg.w[2])
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (y2 + j < sg.h[2]) *(unsigned *)(base + sg.off[2] + (size_t)(y2 + j) * sg.pitch[2] + x2) = l2[j];
    const unsigned t = l2[0] + l2[1];
    const unsigned l3 = (t & 0xFFFFu) + (t >> 16);
    if (x3 < sg.w[3] && y3 < sg.h[3]) base[sg.off[3] + (size_t)y3 * sg.pitch[3] + x3] = (unsigned short)l3;
    // scale 4: the 2 x 2 lane group (lanes are 16 x 4 row-major: right neighbour lane ^ 1, lower one lane ^ 16)
    const unsigned l4 = (l3 + tm_shfl_xor_u32(l3, 1)) + (tm_shfl_xor_u32(l3, 16) + tm_shfl_xor_u32(l3, 17));
    if (!(lane & 1) && !(lane & 16) && (x0 >> 4) < sg.w[4] && (y0 >> 4) < sg.h[4])
        base[sg.off[4] + (size_t)(y0 >> 4) * sg.pitch[4] + (x0 >> 4)] = (unsigned short)l4;
}

// n / d for operands far from the ends of the exponent range (here: |n| <= 2.7e5 or 0, 6.5 <= d <= 2.7e5): the sequence the
// compiler emits for an IEEE division -- reciprocal, one Newton step on it, quotient, two residual corrections -- without
// v_div_scale / v_div_fixup, which only act on operands that need rescaling or are special: same operations on the same
// values, hence the same correctly rounded quotient, 8 instead of 12 instructions.
__device__ __forceinline__ float ssim_div(float n, float d) { return tm_div_inrange(n, d); } // (tm_platform.h)

// one image column of one row on its way through LDS: the sample pair and its two per-sample quantities
struct __attribute__((aligned(16))) TmSsimCol { tmdev::tm_f2 rd, sp; }; // {ref, dis}, {ref^2 + dis^2, ref * dis}

// One (strip, segment) item.  S0: scale 0 (u8 planes) / pyramid scale (u16 box sums, value = sum * inv); NEED_L: also the
// luminance term and the sum of l * cs.  acc: [sum of l * cs, sum of cs] of the lane's valid windows.
template <bool S0, bool NEED_L>
__device__ __forceinline__ void ssim_strip(TmSsimCol *__restrict__ bufE, TmSsimCol *__restrict__ bufO, const void *__restrict__ pr, const void *__restrict__ pd,
                                           int pitch, float inv, int w, int h, int x, int y_base, int y_end, const float (&gw)[TM_SSIM_TAPS],
                                           double (&acc)[2])
{
    using tmdev::tm_f2;
    using tmdev::f2_fma;
    using tmdev::f2_make;
    using tmdev::f2_splat;
    const int lane = threadIdx.x & 63;
    const bool in0 = x < w, in1 = x + 1 < w;
    // the lane's two samples of row y, both sides, as raw integers (rows past the image: 0)
    auto load = [&](int y, unsigned &a, unsigned &b) {
        const int yc = y < h ? y : h - 1;
        // x is even and the pitch a multiple of 64 elements: the pair load stays inside the (padded) row; lanes past the image load
        // column 0 (no branch around the loads: the compiler then counts them exactly, s_waitcnt vmcnt(2 (PF - 1)))
        // the row's address is wave-uniform: pinned into SGPRs (scalar multiply) so that the lane part is a 32-bit offset -- left
        // to the compiler it is a 64-bit vector multiply-add (quarter rate) per load
        const size_t ro = (size_t)yc * (size_t)pitch * (S0 ? 1 : 2);
        TM_GLOBAL_AS const char *ra = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)pr + ro);
        TM_GLOBAL_AS const char *rb = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)pd + ro);
        const unsigned xo = in0 ? (unsigned)x * (S0 ? 1u : 2u) : 0u;
        unsigned va, vb;
        if (S0) { va = *(TM_GLOBAL_AS const unsigned short *)(ra + xo); vb = *(TM_GLOBAL_AS const unsigned short *)(rb + xo); }
        else { va = *(TM_GLOBAL_AS const unsigned *)(ra + xo); vb = *(TM_GLOBAL_AS const unsigned *)(rb + xo); }
        a = va; b = vb;
        (void)y;
    };
    auto unpack = [&](unsigned raw, int y, float &v0, float &v1) { // samples of row y; rows and columns past the image read as 0
        if (S0) { v0 = (float)(raw & 255u); v1 = (float)((raw >> 8) & 255u); }
        else { v0 = (float)(raw & 0xFFFFu) * inv; v1 = (float)(raw >> 16) * inv; }
        const bool ok = y < h;
        if (!(ok && in0)) v0 = 0.0f;
        if (!(ok && in1)) v1 = 0.0f;
    };
    // Rows of load prefetch: a RING with static slots (slot = row % PF).  The step loop is unrolled by WS = 12 -- the window of
    // row-filtered values below holds one row more than the 11 it needs, so that 12 % PF == 0 -- and no register that a load is
    // still writing is ever moved: round 2's shift register (pa[k] = pa[k + 1]) made every step wait for the load of the step
    // before (s_waitcnt vmcnt(0) ~180 instructions after its issue), i.e. a prefetch distance of ONE row whatever PF said.
    constexpr int PF = 3, WS = 12;
    static_assert(WS % PF == 0 && WS > TM_SSIM_TAPS, "ring and window sizes");
    unsigned pa[PF], pb[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) load(y_base + k, pa[k], pb[k]);
    tm_f2 g2[TM_SSIM_TAPS];
#pragma unroll
    for (int k = 0; k < TM_SSIM_TAPS; ++k) g2[k] = f2_splat(gw[k]);
    // the window of row-filtered values (WS slots, the newest 11 rows are read): per column the {x, y} and the {x^2 + y^2, xy} pair
    tm_f2 w01[WS][2], w23[WS][2];
#pragma unroll
    for (int k = 0; k < WS; ++k) { w01[k][0] = w01[k][1] = w23[k][0] = w23[k][1] = f2_splat(0.0f); }
    const tm_f2 C1 = f2_splat(6.5025f), C2 = f2_splat(58.5225f), two = f2_splat(2.0f); // (0.01*255)^2, (0.03*255)^2
    double a_l[2] = {0.0, 0.0}, a_cs[2] = {0.0, 0.0};
    const int n_rows = (y_end - y_base) + 10; // input rows y_base .. y_end+9
    // whole groups of WS steps (up to WS - 1 steps past the segment's last input row: their windows are not accumulated), so that the
    // step body has no exit and its loads and waits are counted exactly
    for (int t0 = 0; t0 < n_rows; t0 += WS) {
#pragma unroll
        for (int j = 0; j < WS; ++j) {
            const int t = t0 + j;
            {
                float r0, r1, d0, d1;
                unpack(pa[j % PF], y_base + t, r0, r1); // row t (requested PF steps ago)
                unpack(pb[j % PF], y_base + t, d0, d1);
                load(y_base + t + PF, pa[j % PF], pb[j % PF]);
                // columns x .. x+11 as {ref, dis} and {ref^2 + dis^2, ref * dis} pairs: the own two computed here, ten from the neighbours
                TmSsimCol c[12];
                c[0].rd = f2_make(r0, d0); c[0].sp = f2_make(__builtin_fmaf(r0, r0, d0 * d0), r0 * d0);
                c[1].rd = f2_make(r1, d1); c[1].sp = f2_make(__builtin_fmaf(r1, r1, d1 * d1), r1 * d1);
                __builtin_amdgcn_wave_barrier();
                bufE[lane] = c[0]; bufO[lane] = c[1];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 1; i < 6; ++i) { c[2 * i] = bufE[lane + i]; c[2 * i + 1] = bufO[lane + i]; }
                // row filter of the two windows that start at columns x and x+1: taps ascending, fma from 0
                tm_f2 a01[2] = {f2_splat(0.0f), f2_splat(0.0f)}, a23[2] = {f2_splat(0.0f), f2_splat(0.0f)};
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if (i < 11) { a01[0] = f2_fma(g2[i], c[i].rd, a01[0]); a23[0] = f2_fma(g2[i], c[i].sp, a23[0]); }
                    if (i > 0) { a01[1] = f2_fma(g2[i - 1], c[i].rd, a01[1]); a23[1] = f2_fma(g2[i - 1], c[i].sp, a23[1]); }
                }
                w01[j][0] = a01[0]; w01[j][1] = a01[1]; w23[j][0] = a23[0]; w23[j][1] = a23[1];
                if (t >= 10 && t < n_rows) { // window rows t-10 .. t are in slots (j+2)%WS .. j (wave-uniform test)
                    tm_f2 v01[2] = {f2_splat(0.0f), f2_splat(0.0f)}, v23[2] = {f2_splat(0.0f), f2_splat(0.0f)};
#pragma unroll
                    for (int k = 0; k < TM_SSIM_TAPS; ++k) {
                        const int q = (j + WS - 10 + k) % WS;
                        v01[0] = f2_fma(g2[k], w01[q][0], v01[0]); v01[1] = f2_fma(g2[k], w01[q][1], v01[1]);
                        v23[0] = f2_fma(g2[k], w23[q][0], v23[0]); v23[1] = f2_fma(g2[k], w23[q][1], v23[1]);
                    }
                    // the two windows side by side: {col 0, col 1} pairs
                    const tm_f2 mx = f2_make(v01[0].x, v01[1].x), my = f2_make(v01[0].y, v01[1].y);
                    const tm_f2 mxx = mx * mx, myy = my * my, mxy = mx * my, mm = mxx + myy;
                    const tm_f2 sv = f2_make(v23[0].x, v23[1].x) - mm, sxy = f2_make(v23[0].y, v23[1].y) - mxy; // sigma_x^2 + sigma_y^2, sigma_xy
                    const tm_f2 csn = f2_fma(two, sxy, C2), csd = sv + C2;
                    const float cs0 = ssim_div(csn.x, csd.x), cs1 = ssim_div(csn.y, csd.y);
                    a_cs[0] += (double)cs0; a_cs[1] += (double)cs1;
                    if (NEED_L) {
                        const tm_f2 ln = f2_fma(two, mxy, C1), ld = mm + C1;
                        const float l0 = ssim_div(ln.x, ld.x), l1 = ssim_div(ln.y, ld.y);
                        a_l[0] += (double)(l0 * cs0); a_l[1] += (double)(l1 * cs1);
                    }
                }
            }
        }
    }
    // windows that start inside the strip and fit the image; the others (halo lanes, right edge) are dropped here
    const bool ok0 = 2 * lane < TM_SSIM_STRIP && x < w - 10, ok1 = 2 * lane + 1 < TM_SSIM_STRIP && x + 1 < w - 10;
    acc[0] = (ok0 ? a_l[0] : 0.0) + (ok1 ? a_l[1] : 0.0);
    acc[1] = (ok0 ? a_cs[0] : 0.0) + (ok1 ? a_cs[1] : 0.0);
}

// grid (slots*3, items of the scales [0, nscales)), block 64.  PART[(slot*3+c)][item][2].
// need_l: bit s set = scale s also needs the sum of l * cs (otherwise PART[..][0] is written as 0).
#ifndef TM_SSIM_WAVES
#define TM_SSIM_WAVES 3
#endif
__global__ void __launch_bounds__(64) TM_WAVES_PER_SIMD(TM_SSIM_WAVES) k_ssim_stream(TmSsimGeom sg, int nscales, unsigned need_l, const unsigned char *__restrict__ Q,
                                                                         const unsigned short *__restrict__ PYR, double *__restrict__ PART)
{
    __shared__ TmSsimCol bufE[72], bufO[72]; // [lane (+ 8 to the right)]: the even / the odd column of the lane's pair
    const int lane = threadIdx.x;
    const int slot = blockIdx.x / 3, c = blockIdx.x % 3;
    int s = 0;
#pragma unroll
    for (int i = 1; i < TM_SSIM_SCALES; ++i)
        if (i < nscales && (int)blockIdx.y >= sg.item_off[i]) s = i;
    const int item = (int)blockIdx.y - sg.item_off[s];
    const int w = sg.w[s], h = sg.h[s];
    const int x_base = (item % sg.strips_x[s]) * TM_SSIM_STRIP, y_base = (item / sg.strips_x[s]) * sg.seg_rows[s];
    const int y_end = min(y_base + sg.seg_rows[s], h - 10); // window rows [y_base, y_end)
    const int x = x_base + 2 * lane;
    if (lane < 8) { // the halo lanes' right-hand neighbours
        TmSsimCol z; z.rd = z.sp = tmdev::f2_splat(0.0f);
        bufE[64 + lane] = z; bufO[64 + lane] = z;
    }
    float gw[TM_SSIM_TAPS];
#pragma unroll
    for (int k = 0; k < TM_SSIM_TAPS; ++k) gw[k] = sg.g[k];
    double acc[2] = {0.0, 0.0};
    const bool nl = (need_l >> s) & 1u;
    if (s == 0) {
        const unsigned char *qr = Q + ((size_t)(slot * 2 + 0) * 3 + c) * sg.qplane, *qd = Q + ((size_t)(slot * 2 + 1) * 3 + c) * sg.qplane;
        if (nl) ssim_strip<true, true>(bufE, bufO, qr, qd, sg.pitch[0], 1.0f, w, h, x, y_base, y_end, gw, acc);
        else ssim_strip<true, false>(bufE, bufO, qr, qd, sg.pitch[0], 1.0f, w, h, x, y_base, y_end, gw, acc);
    } else {
        const unsigned short *fr = ssim_plane_s(sg, PYR, slot * 2 + 0, c, s), *fd = ssim_plane_s(sg, PYR, slot * 2 + 1, c, s);
        const float inv = 1.0f / (float)(1 << (2 * s)); // 4^-s, exact
        if (nl) ssim_strip<false, true>(bufE, bufO, fr, fd, sg.pitch[s], inv, w, h, x, y_base, y_end, gw, acc);
        else ssim_strip<false, false>(bufE, bufO, fr, fd, sg.pitch[s], inv, w, h, x, y_base, y_end, gw, acc);
    }
    double a6[6] = {acc[0], acc[1], 0, 0, 0, 0};
    if (tm_wave_sum6(a6)) {
        double *o = PART + (((size_t)slot * 3 + c) * sg.item_off[TM_SSIM_SCALES] + blockIdx.y) * 2;
        o[0] = a6[0]; o[1] = a6[1];
    }
}

// SUMS[slot][channel 3][scale 5][ssim, cs]: lane l adds items l, l+64, ... in order, then the 64 lane totals are added
// by a fixed tree (deterministic run to run); scales that were not run read 0.  grid (slots, 30), block 64.
__global__ void __launch_bounds__(64) k_ssim_finish(TmSsimGeom sg, int nscales, const double *__restrict__ PART, double *__restrict__ SUMS)
{
    const int slot = blockIdx.x, i = blockIdx.y;
    const int c = i / 10, s = (i % 10) / 2, which = i & 1;
    double a[6] = {0, 0, 0, 0, 0, 0};
    if (s < nscales)
        for (int t = sg.item_off[s] + (int)threadIdx.x; t < sg.item_off[s + 1]; t += 64)
            a[0] += PART[(((size_t)slot * 3 + c) * sg.item_off[TM_SSIM_SCALES] + t) * 2 + which];
    if (tm_wave_sum6(a)) SUMS[(size_t)slot * 30 + i] = a[0];
}

} // namespace tmk
