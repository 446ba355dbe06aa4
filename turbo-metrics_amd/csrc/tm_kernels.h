// tm_kernels.h -- gfx950 kernels of the SSIMULACRA2 / PSNR frame-pair path.
//
// Two pipelines live here:
//   default    k_ingest_rows<KIND> (4:2:0 kinds; k_ingest_wave<KIND> for the RGB kinds and mixed launches) + k_ingest_upper_rd
//              -> k_blur_v_jobs<32, 16> -> k_blur_h_jobs_x (k_blur_h_jobs_split, eight waves per row block, for small launches) -> k_finish_jobs
//              over the ref/dis-interleaved XYB pyramid, job-table driven, slot-major grids (x = slot); larger launches send the
//              edge-only jobs through k_blur_edge_fused + k_finish_edge (one kernel, no pass-1 planes) beside the two passes
//   reference  k_ingest + k_downscale + k_xyb -> k_blur_v -> k_blur_h_jobs -> k_finish_jobs: straight-line, LDS-free kernels
//              whose only job is to be obviously correct (engine variant TM_VARIANT_REFERENCE): tm_reference_kernels.h, laboratory
//              build only; the GPU tier checks that the two pipelines produce identical bits on the device, and both against the CPU oracle.
//
// Launch geometry (64-lane wavefront == 1 workgroup unless noted):
//   k_ingest_rows     grid (ceil(ceil(w/2)/64), ceil(ceil(h/2)/(4 rows_per_wave)), slots)  block 256  four waves, each 64 quads x rows_per_wave quad rows of both sides
//   k_ingest_wave     grid (ceil(w/32), ceil(h/8), slots)            block 64    one 32 x 8 tile, both sides
//   k_ingest_upper_rd grid (ceil(w2/32), ceil(h2/32), slots)         block 256   pyramid levels 2..5
//   k_blur_v_jobs     grid (slots, jobs.vstart[n])                   block 320   column pass, five role-waves per 64 columns
//   k_blur_h_jobs_x   grid (slots, jobs.hstart[n])                   block 64    row pass + error maps + sums, lane = image row
//   k_blur_h_jobs_split grid (slots, jobs.hstart[n])                 block 512   the same row pass over eight waves per row block (small launches)
//   k_finish_jobs     grid (slots)                                   block 128
//   k_blur_edge_fused<4, grouped>  grid (tickets = slots * edge jobs * ceil(bands of 32 rows / 4), or fewer: persistent)  block 256   four waves = four adjacent bands of one (slot, job) per ticket
//   k_finish_edge     grid (slots * edge jobs)                       block 64
//
// Arithmetic follows the reference kernels operation for operation (cited per function); the
// file must be compiled with -ffp-contract=off so that only the explicit fmaf calls fuse.
#pragma once
#include "tm_device_math.h"
#include "tm_geom.h"
#include <type_traits>

// (wave-level sum / shuffle helpers, TM_LDS_BARRIER, TM_WAVES_PER_SIMD, tm_mul24, tm_f4 / tm_g2, DPP lane exchanges: tm_platform.h)

namespace tmk {

// row `row` of a plane whose base pointer is wave-uniform: the row address stays in SGPRs and the load uses
// the scalar-base + per-lane-offset form, so a whole window of in-flight loads costs one VGPR of addressing
__device__ __forceinline__ tm_f4 tm_make_f4(float a, float b, float c, float d) { tm_f4 v = {a, b, c, d}; return v; }
template <typename T> __device__ __forceinline__ TM_GLOBAL_AS T *tm_uniform_ptr(T *p)
{
    // Pin a wave-uniform pointer into an SGPR pair (and keep it in the global address space).  The empty asm
    // is opaque to LLVM, which otherwise re-associates base + row*pitch + lane into a per-lane 64-bit address
    // for every load of the window (2 VGPRs and a v_lshl_add_u64 each) instead of selecting the
    // scalar-base + 32-bit-lane-offset form of global_load / global_store.
    unsigned long long v = (unsigned long long)p;
    TM_PIN_SGPR(v);
    return (TM_GLOBAL_AS T *)v;
}

__device__ __forceinline__ float ld_row(const float *__restrict__ p, int row, int nrows, int pitch)
{
    const int rc = row < nrows ? row : nrows - 1;
    const float v = p[(size_t)rc * pitch];
    return row < nrows ? v : 0.0f;
}
// ------------------------------------------------------------------------------------------------
// ingest: decoded frame -> planar linear RGB (scale 0) for both sides of a slot, + integer SSE of
// the u8-quantised pair for PSNR.
//   NV12/P016  cuda-colorspace-kernel/src/biplanar.rs:8-70 (one 2x2 luma quad per lane, like the
//              reference; an odd last column/row is not converted by the reference
//              (cuda-colorspace/src/kernel.rs:64-65) and is written as 0 here)
//   RGB8       srgb.rs:51-66 (LUT);  RGB16/RGBF32 srgb.rs:68-127;  LINEARF32: plain copy
//   quantise   sample_conv.rs:6-35 (float2uint_rn(v*255))
// coef: [matrix 0..2][bits 8|16][5] = y, r, b, g1, g2 coefficients (lib.rs:186-200), host computed.
// ------------------------------------------------------------------------------------------------
//   planar     TM_KIND_I420_8 / I420_16: the same conversion; Cb and Cr come from two planes and a 16-bit sample is first
//              shifted to the top of its 16 bits (a P016 surface holds exactly that)
// sample x of a TM_KIND_I420_P10 row (tm_geom.h): block x / 384, run (x % 384) / 128, word x % 128
__device__ __forceinline__ unsigned p10_word_offset(unsigned x) { return ((x / TM_P10_BLOCK) * TM_P10_RUN + (x % TM_P10_RUN)) * 4u; }
__device__ __forceinline__ unsigned p10_shift(unsigned x) { return 10u * ((x % TM_P10_BLOCK) / TM_P10_RUN); }
__device__ __forceinline__ unsigned p10_sample(const char *row, unsigned x)
{
    return (*(const unsigned *)(row + p10_word_offset(x)) >> p10_shift(x)) & 1023u;
}
template <typename T, int BITS>
__device__ __forceinline__ void ingest_yuv_quad(const TmFrameDesc &d, const float *__restrict__ coef,
                                                const double *__restrict__ tab, int qx, int qy, float (&px)[2][2][3])
{
    const float *k = coef + (d.matrix * 2 + (BITS == 16 ? 1 : 0)) * 5;
    const int neutral = 1 << (BITS - 1);
    const unsigned ymin = 16u << (BITS - 8);
    const bool planar = d.kind == TM_KIND_I420_8 || d.kind == TM_KIND_I420_16 || d.kind == TM_KIND_I420_P10;
    const int sh = planar && BITS == 16 ? d.shift : 0;
    unsigned ucb, ucr, y00, y01, y10, y11;
    if (BITS == 16 && d.kind == TM_KIND_I420_P10) { // three samples per word (tm_geom.h)
        const char *yr0 = (const char *)d.p0 + (size_t)(2 * qy) * d.pitch, *yr1 = yr0 + d.pitch;
        y00 = p10_sample(yr0, 2 * qx); y01 = p10_sample(yr0, 2 * qx + 1); y10 = p10_sample(yr1, 2 * qx); y11 = p10_sample(yr1, 2 * qx + 1);
        ucb = p10_sample((const char *)d.p1 + (size_t)qy * d.pitch2, qx);
        ucr = p10_sample((const char *)d.p2 + (size_t)qy * d.pitch2, qx);
    } else {
        // all six samples of the quad are fetched before any is used: one exposed latency instead of six
        const T *yrow0 = (const T *)((const char *)d.p0 + (size_t)(2 * qy) * d.pitch) + 2 * qx;
        const T *yrow1 = (const T *)((const char *)d.p0 + (size_t)(2 * qy + 1) * d.pitch) + 2 * qx;
        if (planar) {
            ucb = ((const T *)((const char *)d.p1 + (size_t)qy * d.pitch2))[qx];
            ucr = ((const T *)((const char *)d.p2 + (size_t)qy * d.pitch2))[qx];
        } else {
            const T *uv = (const T *)((const char *)d.p1 + (size_t)qy * d.pitch) + 2 * qx;
            ucb = uv[0]; ucr = uv[1];
        }
        y00 = yrow0[0]; y01 = yrow0[1]; y10 = yrow1[0]; y11 = yrow1[1];
    }
    ucb = (ucb << sh) & 0xFFFFu; ucr = (ucr << sh) & 0xFFFFu;
    const unsigned yv[2][2] = {{(y00 << sh) & 0xFFFFu, (y01 << sh) & 0xFFFFu}, {(y10 << sh) & 0xFFFFu, (y11 << sh) & 0xFFFFu}};
    const float cb = (float)((int)ucb - neutral);
    const float cr = (float)((int)ucr - neutral);
    const float r_ = k[1] * cr;
    const float g_ = __builtin_fmaf(k[3], cb, k[4] * cr);
    const float b_ = k[2] * cb;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const unsigned ys = yv[iy][ix];
            const float luma = (float)((ys > ymin ? ys : ymin) - ymin) * k[0];
            px[iy][ix][0] = tmdev::clamp01(tmdev::bt709_eotf(luma + r_, tab));
            px[iy][ix][1] = tmdev::clamp01(tmdev::bt709_eotf(luma + g_, tab));
            px[iy][ix][2] = tmdev::clamp01(tmdev::bt709_eotf(luma + b_, tab));
        }
    }
}
// The six samples of a quad as three pair loads (two luma rows, one CbCr pair): half the load instructions and half the
// registers, which is what lets k_ingest_wave hold BOTH sides' samples from the start.  raw[i] = first | second << bits.
// Pair loads need the plane pointers and the pitch to be multiples of the pair size; otherwise single loads are packed.
// PLANAR: Cb and Cr are single loads from their own planes (16 lanes along x = 16 / 32 contiguous bytes of each), and 16-bit
// samples are shifted to the top of their half (masked first: bits above the declared depth cannot spill into the neighbour).
template <typename T, bool PLANAR>
__device__ __forceinline__ void yuv_quad_load_pairs(const TmFrameDesc &d, int qx, int qy, unsigned (&raw)[3])
{
    const char *y0 = (const char *)d.p0 + tm_mul24((unsigned)(2 * qy), (unsigned)d.pitch) + (unsigned)(2 * qx) * (unsigned)sizeof(T);
    const char *y1 = y0 + d.pitch;
    const int sh = 8 * (int)sizeof(T);
    const bool aligned = (((unsigned long long)d.p0 | (PLANAR ? 0ull : (unsigned long long)d.p1) | (unsigned long long)d.pitch) & (2 * sizeof(T) - 1)) == 0; // wave-uniform
    if (aligned) {
        if (sizeof(T) == 1) { raw[0] = *(const unsigned short *)y0; raw[1] = *(const unsigned short *)y1; }
        else { raw[0] = *(const unsigned *)y0; raw[1] = *(const unsigned *)y1; }
    } else {
        raw[0] = (unsigned)((const T *)y0)[0] | ((unsigned)((const T *)y0)[1] << sh);
        raw[1] = (unsigned)((const T *)y1)[0] | ((unsigned)((const T *)y1)[1] << sh);
    }
    if (PLANAR) {
        const unsigned coff = tm_mul24((unsigned)qy, (unsigned)d.pitch2) + (unsigned)qx * (unsigned)sizeof(T);
        const unsigned cb = *(const T *)((const char *)d.p1 + coff);
        const unsigned cr = *(const T *)((const char *)d.p2 + coff);
        raw[2] = cb | (cr << sh);
        if (sizeof(T) == 2) {
            const unsigned keep = (0xFFFFu >> d.shift) * 0x10001u;
#pragma unroll
            for (int i = 0; i < 3; ++i) raw[i] = (raw[i] & keep) << d.shift;
        }
    } else {
        const char *uv = (const char *)d.p1 + tm_mul24((unsigned)qy, (unsigned)d.pitch) + (unsigned)(2 * qx) * (unsigned)sizeof(T);
        if (aligned) raw[2] = sizeof(T) == 1 ? (unsigned)*(const unsigned short *)uv : *(const unsigned *)uv;
        else raw[2] = (unsigned)((const T *)uv)[0] | ((unsigned)((const T *)uv)[1] << sh);
    }
}
// TM_KIND_I420_P10: the same three raw words -- first | second << 16, each sample shifted to the top of its half like I420_16 with
// shift 6 -- out of the packed rows: the two luma samples of a quad row are neighbouring words of one run (2 qx is even, a run has 128
// words: one aligned 8-byte load), behind one shift
__device__ __forceinline__ void yuv_quad_load_p10(const TmFrameDesc &d, int qx, int qy, unsigned (&raw)[3])
{
    const unsigned x = (unsigned)(2 * qx), oy = p10_word_offset(x), sy = p10_shift(x), oc = p10_word_offset((unsigned)qx), sc = p10_shift((unsigned)qx);
    const char *y0 = (const char *)d.p0 + tm_mul24((unsigned)(2 * qy), (unsigned)d.pitch) + oy;
    const tm_u2 a = *(const tm_u2 *)y0, b = *(const tm_u2 *)(y0 + d.pitch);
    const unsigned coff = tm_mul24((unsigned)qy, (unsigned)d.pitch2) + oc;
    const unsigned cb = *(const unsigned *)((const char *)d.p1 + coff), cr = *(const unsigned *)((const char *)d.p2 + coff);
    raw[0] = ((((a.x >> sy) & 1023u) | (((a.y >> sy) & 1023u) << 16)) << 6);
    raw[1] = ((((b.x >> sy) & 1023u) | (((b.y >> sy) & 1023u) << 16)) << 6);
    raw[2] = ((((cb >> sc) & 1023u) | (((cr >> sc) & 1023u) << 16)) << 6);
}
// ... and for a wave of k_ingest_rows (64 quads from x = 128 blockIdx.x: one run of the luma rows, half a run of the chroma rows): row
// addresses in SGPRs, the lane's byte offsets xo_y / xo_c fixed for the whole walk, both shifts wave-uniform
__device__ __forceinline__ void yuv_row_load_p10(const TmFrameDesc &d, unsigned xo_y, unsigned xo_c, unsigned sy, unsigned sc, int qy, unsigned (&raw)[3])
{
    TM_GLOBAL_AS const char *y0 = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p0 + (size_t)(2 * qy) * d.pitch);
    TM_GLOBAL_AS const char *y1 = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p0 + (size_t)(2 * qy + 1) * d.pitch);
    TM_GLOBAL_AS const char *cbp = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p1 + (size_t)qy * d.pitch2);
    TM_GLOBAL_AS const char *crp = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p2 + (size_t)qy * d.pitch2);
    const tm_u2 a = *(TM_GLOBAL_AS const tm_u2 *)(y0 + xo_y), b = *(TM_GLOBAL_AS const tm_u2 *)(y1 + xo_y);
    const unsigned cb = *(TM_GLOBAL_AS const unsigned *)(cbp + xo_c), cr = *(TM_GLOBAL_AS const unsigned *)(crp + xo_c);
    raw[0] = ((((a.x >> sy) & 1023u) | (((a.y >> sy) & 1023u) << 16)) << 6);
    raw[1] = ((((b.x >> sy) & 1023u) | (((b.y >> sy) & 1023u) << 16)) << 6);
    raw[2] = ((((cb >> sc) & 1023u) | (((cr >> sc) & 1023u) << 16)) << 6);
}
// The same six samples for a wave whose lanes all sit in ONE quad row (k_ingest_rows: qy is wave-uniform): the row addresses stay in
// SGPRs and a lane contributes a 32-bit byte offset that does not change from row to row (xo_y into a luma row, xo_c into a chroma
// row) -- scalar-base + lane-offset global loads.  With per-lane 64-bit addresses (yuv_quad_load_pairs) the lane bases were spilled
// around the row loop, and the reload's s_waitcnt vmcnt(0) drained the previous row's fifteen stores before every prefetch.
template <typename T, bool PLANAR>
__device__ __forceinline__ void yuv_row_load_pairs(const TmFrameDesc &d, unsigned xo_y, unsigned xo_c, int qy, unsigned (&raw)[3])
{
    const int sh = 8 * (int)sizeof(T);
    const bool aligned = (((unsigned long long)d.p0 | (PLANAR ? 0ull : (unsigned long long)d.p1) | (unsigned long long)d.pitch) & (2 * sizeof(T) - 1)) == 0; // wave-uniform
    TM_GLOBAL_AS const char *y0 = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p0 + (size_t)(2 * qy) * d.pitch);
    TM_GLOBAL_AS const char *y1 = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p0 + (size_t)(2 * qy + 1) * d.pitch);
    if (aligned) {
        if (sizeof(T) == 1) { raw[0] = *(TM_GLOBAL_AS const unsigned short *)(y0 + xo_y); raw[1] = *(TM_GLOBAL_AS const unsigned short *)(y1 + xo_y); }
        else { raw[0] = *(TM_GLOBAL_AS const unsigned *)(y0 + xo_y); raw[1] = *(TM_GLOBAL_AS const unsigned *)(y1 + xo_y); }
    } else {
        raw[0] = (unsigned)((TM_GLOBAL_AS const T *)(y0 + xo_y))[0] | ((unsigned)((TM_GLOBAL_AS const T *)(y0 + xo_y))[1] << sh);
        raw[1] = (unsigned)((TM_GLOBAL_AS const T *)(y1 + xo_y))[0] | ((unsigned)((TM_GLOBAL_AS const T *)(y1 + xo_y))[1] << sh);
    }
    if (PLANAR) {
        TM_GLOBAL_AS const char *cbp = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p1 + (size_t)qy * d.pitch2);
        TM_GLOBAL_AS const char *crp = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p2 + (size_t)qy * d.pitch2);
        const unsigned cb = *(TM_GLOBAL_AS const T *)(cbp + xo_c);
        const unsigned cr = *(TM_GLOBAL_AS const T *)(crp + xo_c);
        raw[2] = cb | (cr << sh);
        if (sizeof(T) == 2) {
            const unsigned keep = (0xFFFFu >> d.shift) * 0x10001u;
#pragma unroll
            for (int i = 0; i < 3; ++i) raw[i] = (raw[i] & keep) << d.shift;
        }
    } else {
        TM_GLOBAL_AS const char *uv = (TM_GLOBAL_AS const char *)tm_uniform_ptr((const char *)d.p1 + (size_t)qy * d.pitch);
        if (aligned) raw[2] = sizeof(T) == 1 ? (unsigned)*(TM_GLOBAL_AS const unsigned short *)(uv + xo_y) : *(TM_GLOBAL_AS const unsigned *)(uv + xo_y);
        else raw[2] = (unsigned)((TM_GLOBAL_AS const T *)(uv + xo_y))[0] | ((unsigned)((TM_GLOBAL_AS const T *)(uv + xo_y))[1] << sh);
    }
}
template <int BITS> __device__ __forceinline__ void yuv_quad_unpack(const unsigned (&pr)[3], unsigned (&raw)[6])
{
    const unsigned m = BITS == 8 ? 0xFFu : 0xFFFFu;
    raw[0] = pr[0] & m; raw[1] = pr[0] >> BITS; raw[2] = pr[1] & m; raw[3] = pr[1] >> BITS; raw[4] = pr[2] & m; raw[5] = pr[2] >> BITS;
}

template <int BITS>
__device__ __forceinline__ void yuv_quad_convert(const TmFrameDesc &d, const unsigned (&raw)[6], const float *__restrict__ coef,
                                                 const double *__restrict__ tab, float (&px)[2][2][3])
{
    const float *k = coef + (d.matrix * 2 + (BITS == 16 ? 1 : 0)) * 5;
    const int neutral = 1 << (BITS - 1);
    const unsigned ymin = 16u << (BITS - 8);
    const unsigned ucb = raw[4], ucr = raw[5];
    const float cb = (float)((int)ucb - neutral);
    const float cr = (float)((int)ucr - neutral);
    const float r_ = k[1] * cr;
    const float g_ = __builtin_fmaf(k[3], cb, k[4] * cr);
    const float b_ = k[2] * cb;
    // (memoising R and B of 8-bit frames in two 256 x 256 tables per matrix, and a 10-bit variant for P016, were measured: the
    // gathers cost more than the evaluations they replace; docs/LABBOOK.md section 5.1)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned ys = raw[q];
        const float luma = (float)((ys > ymin ? ys : ymin) - ymin) * k[0];
        px[q >> 1][q & 1][0] = tmdev::clamp01(tmdev::bt709_eotf(luma + r_, tab));
        px[q >> 1][q & 1][1] = tmdev::clamp01(tmdev::bt709_eotf(luma + g_, tab));
        px[q >> 1][q & 1][2] = tmdev::clamp01(tmdev::bt709_eotf(luma + b_, tab));
    }
}

__device__ __forceinline__ float ds4(float v00, float v01, float v10, float v11, bool okx, bool oky)
{
    // sum order of downscale.rs:22-30: (iy,ix) = (0,0),(0,1),(1,0),(1,1), starting from 0.0
    const float b = okx ? v01 : v00;
    const float c = oky ? v10 : v00;
    const float d = okx ? (oky ? v11 : v01) : (oky ? v10 : v00);
    float sum = 0.0f;
    sum += v00; sum += b; sum += c; sum += d;
    return sum * 0.25f;
}

__device__ __forceinline__ void ingest_px_rgb(const TmFrameDesc &d, int kind, const float *__restrict__ lut,
                                              const double *__restrict__ tab, int x, int y, float (&v)[3])
{
    const char *row = (const char *)d.p0 + (size_t)y * d.pitch;
    if (kind == TM_KIND_RGB8) { // all three samples first, then the LUT gathers: one wait instead of three
        const unsigned char *p = (const unsigned char *)row + 3 * x;
        const unsigned a = p[0], b = p[1], c = p[2];
        v[0] = lut[a]; v[1] = lut[b]; v[2] = lut[c];
    } else if (kind == TM_KIND_RGB16) {
        const unsigned short *p = (const unsigned short *)row + 3 * x;
        const float a = (float)p[0], b = (float)p[1], c = (float)p[2];
        v[0] = tmdev::srgb_inverse_oetf(a / 65535.0f, tab);
        v[1] = tmdev::srgb_inverse_oetf(b / 65535.0f, tab);
        v[2] = tmdev::srgb_inverse_oetf(c / 65535.0f, tab);
    } else {
        const float *p = (const float *)row + 3 * x;
        const float a = p[0], b = p[1], c = p[2];
        if (kind == TM_KIND_RGBF32) {
            v[0] = tmdev::srgb_inverse_oetf(a, tab); v[1] = tmdev::srgb_inverse_oetf(b, tab); v[2] = tmdev::srgb_inverse_oetf(c, tab);
        } else { v[0] = a; v[1] = b; v[2] = c; }
    }
}

// ------------------------------------------------------------------------------------------------
// Ingest ("wave"): decoded frame pair -> linear RGB -> XYB of pyramid levels 0 and 1 + the level-2 LINEAR pixels, no LDS
// tile, no barrier.  One wave = one 32 x 8 pixel tile (16 x 4 quads, lane = quad = the unit the reference's NV12 kernel
// works on), both sides one after the other (so the PSNR SSE needs no second pass).
// The XYB pyramid holds ref and dis in ONE plane, interleaved per pixel ([y][x][side], rows of 2 * pitch floats): both blur
// passes need both sides of a pixel at the same time and fetch them with one load of whole 128-B lines.  Side 0's XYB
// waits in wave-private LDS; when side 1 is done the lane stores {ref, dis} pairs straight from registers: a float4 per
// row (two pixels x two sides), 16 lanes along x = 256 contiguous bytes.  The level-2 linear pixels (8 x 2 per tile) come
// from the level-1 pixels of four neighbouring lanes through wave shuffles and go to LIN2 for k_ingest_upper_rd.
// KIND >= 0: every frame of the launch has this TM_KIND_* (the normal case; the host checks), so all format branches fold
// away and the sample loads of both sides are issued back to back; KIND = -1: per-frame dispatch.
// grid (ceil(w/32), ceil(h/8), slots), block 64.
// ------------------------------------------------------------------------------------------------
// (KIND = -1 carries every format's code: held to the 96 registers of five waves per SIMD it spilled 18 of them -- 76 bytes of scratch per lane;
// the mixed launch is the rare path and runs at four waves per SIMD instead)
template <int KIND>
__global__ void __launch_bounds__(64) TM_WAVES_PER_SIMD(KIND < 0 ? 4 : 5) k_ingest_wave(TmGeom g, const TmFrameDesc *__restrict__ desc, const float *__restrict__ lut,
                                                    const float *__restrict__ coef, const double *__restrict__ gtab,
                                                    float *__restrict__ XYB, float *__restrict__ LIN2,
                                                    unsigned long long *__restrict__ SSE, int want_sse,
                                                    unsigned char *__restrict__ QU8, unsigned long long qplane, int qpitch)
{
    __shared__ double tab[TM_TAB_POW_DOUBLES]; // pow_pos tables (RGB16 / RGBF32); the transfer-function table of the YUV kinds is read from
    const double *__restrict__ et64 = gtab + TM_TAB_EOTF64; // global memory here (16 KB, cache resident): this kernel is the fallback of the 4:2:0 kinds
    const int lane = threadIdx.x;
    const int qx = lane & 15, qy = lane >> 4;
    const int slot = blockIdx.z;
    // (an XCD-aware tile order -- the four tiles that share a 128-B line of an NV12 row sent to the same XCD's L2 -- measured
    // 8 % slower: four L2s fetching the line in parallel beat one fetching it once; docs/LABBOOK.md section 5.1)
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 8;
    const int X0 = tx0 + 2 * qx, Y0 = ty0 + 2 * qy;
    const int w = g.s[0].w, h = g.s[0].h;
    unsigned qref[3] = {0, 0, 0};
    unsigned sse3[3] = {0, 0, 0};
    for (int i = lane; i < TM_TAB_POW_DOUBLES; i += 64) tab[i] = gtab[i];
    __builtin_amdgcn_wave_barrier();
    // side 0's XYB (four level-0 pixels + the level-1 pixel) waits in LDS until side 1 is done ([value][lane]: wave-private,
    // conflict-free; holding it in 15 VGPRs cost the fifth wave per SIMD)
    __shared__ float keep_s[15][64];
    // YUV kinds: the samples of BOTH sides are requested before anything else (three pair loads per side), so that side 1's
    // never sit behind side 0's arithmetic
    constexpr bool YUV = KIND == TM_KIND_NV12 || KIND == TM_KIND_P016 || KIND == TM_KIND_I420_8 || KIND == TM_KIND_I420_16 || KIND == TM_KIND_I420_P10;
    constexpr bool PLANAR = KIND == TM_KIND_I420_8 || KIND == TM_KIND_I420_16;
    constexpr bool YUV8 = KIND == TM_KIND_NV12 || KIND == TM_KIND_I420_8;
    const TmFrameDesc dd0 = desc[slot * 2], dd1 = desc[slot * 2 + 1];
    const bool quad_ok = X0 + 1 < w && Y0 + 1 < h; // incomplete quads are not converted (cuda-colorspace/src/kernel.rs:64-65)
    unsigned pr0[3] = {0, 0, 0}, pr1[3] = {0, 0, 0};
    if (YUV && quad_ok) {
        if (KIND == TM_KIND_I420_P10) { yuv_quad_load_p10(dd0, X0 / 2, Y0 / 2, pr0); yuv_quad_load_p10(dd1, X0 / 2, Y0 / 2, pr1); }
        else if (YUV8) { yuv_quad_load_pairs<unsigned char, PLANAR>(dd0, X0 / 2, Y0 / 2, pr0); yuv_quad_load_pairs<unsigned char, PLANAR>(dd1, X0 / 2, Y0 / 2, pr1); }
        else { yuv_quad_load_pairs<unsigned short, PLANAR>(dd0, X0 / 2, Y0 / 2, pr0); yuv_quad_load_pairs<unsigned short, PLANAR>(dd1, X0 / 2, Y0 / 2, pr1); }
    }
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
        const TmFrameDesc d = side ? dd1 : dd0;
        const int kind = KIND >= 0 ? KIND : d.kind;
        float px[2][2][3];
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                for (int c = 0; c < 3; ++c) px[iy][ix][c] = 0.0f;
        if (YUV) {
            if (quad_ok) {
                const unsigned prs[3] = {side ? pr1[0] : pr0[0], side ? pr1[1] : pr0[1], side ? pr1[2] : pr0[2]};
                unsigned raw[6];
                yuv_quad_unpack<YUV8 ? 8 : 16>(prs, raw);
                if (YUV8) yuv_quad_convert<8>(d, raw, coef, et64, px);
                else yuv_quad_convert<16>(d, raw, coef, et64, px);
            }
        } else if (kind == TM_KIND_NV12 || kind == TM_KIND_P016 || kind == TM_KIND_I420_8 || kind == TM_KIND_I420_16 || kind == TM_KIND_I420_P10) {
            if (quad_ok) {
                if (kind == TM_KIND_NV12 || kind == TM_KIND_I420_8) ingest_yuv_quad<unsigned char, 8>(d, coef, et64, X0 / 2, Y0 / 2, px);
                else ingest_yuv_quad<unsigned short, 16>(d, coef, et64, X0 / 2, Y0 / 2, px);
            }
        } else {
#pragma unroll
            for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                for (int ix = 0; ix < 2; ++ix)
                    if (X0 + ix < w && Y0 + iy < h) ingest_px_rgb(d, kind, lut, tab, X0 + ix, Y0 + iy, px[iy][ix]);
        }
        if (want_sse) { // sample_conv.rs:6-35 quantisation; out-of-image samples are 0 on both sides
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const unsigned q = (unsigned)(int)rintf(px[k / 6][(k / 3) & 1][k % 3] * 255.0f) & 255u;
                if (side == 0) qref[k >> 2] |= q << (8 * (k & 3));
                else {
                    const int dlt = (int)((qref[k >> 2] >> (8 * (k & 3))) & 255u) - (int)q;
                    sse3[k % 3] += (unsigned)(dlt * dlt);
                }
            }
        }
        if (QU8 != nullptr && X0 < w) { // u8-quantised planes for SSIM / MS-SSIM: two pixels = one 16-bit store per row
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
                    if (Y0 + iy < h) {
                        const unsigned q0 = (unsigned)(int)rintf(px[iy][0][c] * 255.0f) & 255u;
                        const unsigned q1 = (unsigned)(int)rintf(px[iy][1][c] * 255.0f) & 255u;
                        *(unsigned short *)(QU8 + ((size_t)(slot * 2 + side) * 3 + c) * qplane + (tm_mul24((unsigned)(Y0 + iy), (unsigned)qpitch) + (unsigned)X0)) = (unsigned short)(q0 | (q1 << 8));
                    }
        }
        if (XYB == nullptr) continue; // PSNR / SSIM only: no pyramid (wave-uniform)
        const bool okx = X0 + 1 < w, oky = Y0 + 1 < h;
        float lr[5], lg[5], lb[5], xa[5], xb[5], xc[5];
#pragma unroll
        for (int k = 0; k < 4; ++k) { lr[k] = px[k >> 1][k & 1][0]; lg[k] = px[k >> 1][k & 1][1]; lb[k] = px[k >> 1][k & 1][2]; }
        lr[4] = ds4(px[0][0][0], px[0][1][0], px[1][0][0], px[1][1][0], okx, oky);
        lg[4] = ds4(px[0][0][1], px[0][1][1], px[1][0][1], px[1][1][1], okx, oky);
        lb[4] = ds4(px[0][0][2], px[0][1][2], px[1][0][2], px[1][1][2], okx, oky);
        // 8- and 16-bit kinds give linear RGB in [0, 1] (clamp01 / the sRGB tables): the cube roots need no range test
        constexpr bool UNIT = YUV || KIND == TM_KIND_RGB8 || KIND == TM_KIND_RGB16;
        tmdev::linear_to_xyb_n<5, UNIT>(lr, lg, lb, xa, xb, xc);
        // ---- level 0: two rows of two pixels; level 1: one pixel
        if (side == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) { keep_s[k][lane] = xa[k]; keep_s[5 + k][lane] = xb[k]; keep_s[10 + k][lane] = xc[k]; }
        } else {
            const TmScaleGeom s0 = g.s[0], s1 = g.s[1];
            const float *xv[3] = {xa, xb, xc};
            float *xi = XYB + (size_t)slot * 2 * g.pyr; // the slot's interleaved pyramid (same size as its two plain ones)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float keep[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) keep[k] = keep_s[5 * c + k][lane];
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
                    if (X0 < w && Y0 + iy < h) // X0 is even and the pitch a multiple of 64 floats: the pair of pixels stays inside the row
                        *(float4 *)(xi + 2 * (s0.off + c * s0.plane) + 2u * (tm_mul24((unsigned)(Y0 + iy), (unsigned)s0.pitch) + (unsigned)X0)) =
                            make_float4(keep[2 * iy], xv[c][2 * iy], keep[2 * iy + 1], xv[c][2 * iy + 1]);
                if (X0 / 2 < s1.w && Y0 / 2 < s1.h)
                    *(float2 *)(xi + 2 * (s1.off + c * s1.plane) + 2u * (tm_mul24((unsigned)(Y0 / 2), (unsigned)s1.pitch) + (unsigned)(X0 / 2))) = make_float2(keep[4], xv[c][4]);
            }
        }
        // ---- level-2 linear pixel of the 2 x 2 lane group (levels 2..5 are finished by k_ingest_upper_rd)
        {
            const TmScaleGeom s1 = g.s[1], s2 = g.s[2];
            const int XL = X0 >> 2, YL = Y0 >> 2;
            const bool ok2x = 2 * XL + 1 < s1.w, ok2y = 2 * YL + 1 < s1.h;
            float v[3];
            const float l1[3] = {lr[4], lg[4], lb[4]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // lanes are (qy, qx) row-major 4 x 16: the right neighbour is lane ^ 1, the lower one lane ^ 16
                const float v01 = tm_shfl_xor(l1[c], 1), v10 = tm_shfl_xor(l1[c], 16), v11 = tm_shfl_xor(l1[c], 17);
                v[c] = ds4(l1[c], v01, v10, v11, ok2x, ok2y);
            }
            if (!(qx & 1) && !(qy & 1) && XL < s2.w && YL < s2.h) {
                float *l2 = LIN2 + (size_t)(slot * 2 + side) * 3 * s2.plane + (tm_mul24((unsigned)YL, (unsigned)s2.pitch) + (unsigned)XL);
#pragma unroll
                for (int c = 0; c < 3; ++c) l2[c * s2.plane] = v[c];
            }
        }
    }
    if (want_sse) {
        if (tm_wave_sum_u32x3(sse3)) {
            const unsigned bin = (blockIdx.x + blockIdx.y * 29) % TM_SSE_BINS;
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicAdd(&SSE[((size_t)slot * TM_SSE_BINS + bin) * 3 + c], (unsigned long long)sse3[c]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ingest ("rows"), the YUV kinds: same results as k_ingest_wave<KIND>, bit for bit, for a launch whose frames are all of one
// 4:2:0 kind.  Written for issue slots, not for arithmetic: on gfx950 one wave issues at most one VALU instruction per four
// cycles while the SIMD retires a 64-lane f32 instruction in two (a packed one in four), so a kernel of dependent scalar-lane
// chains with branches around every transfer-function evaluation (k_ingest_wave: 1 395 VALU + ~700 scalar / branch / wait
// instructions per tile, 0.22 VALU instructions per SIMD-cycle of the 0.5 the SIMD retires) is bound by what its five waves
// can issue.  Here
//   * a lane owns one 2 x 2 quad of BOTH frames and everything is said on {ref, dis} pairs (tm_f2): the conversion, the
//     transfer function, the box filter, the colour mix, the cube roots and the XYB affine steps are v_pk_*_f32 -- half the
//     instructions for the same arithmetic -- and a row of the interleaved pyramid leaves as one float4 per lane straight
//     from the registers the arithmetic produced (no LDS parking, no second pass over side 0's values);
//   * a wave covers 64 quads along x = 128 pixels: one whole 128-B line of an NV12 luma row and of its CbCr row per load
//     (k_ingest_wave's 32-pixel tiles made four workgroups, on different XCDs, fetch every line: 1.59 GB read per 64 1080p
//     pairs for 0.40 GB of frames), 1-KB runs of the pyramid per store;
//   * it walks down `rows_per_wave` quad rows: the samples of the next row are requested before the arithmetic of this one
//     (a wave hides its own load latency), the transfer-function table (16.4 KB) is staged once per four waves x rows_per_wave rows, row addresses advance in SGPRs, and the level-2 linear pixel (2 x 2 level-1 pixels) takes its upper pair from
//     the previous iteration's registers and its right-hand column from lane ^ 1 (one DPP move);
//   * no divergent branch: the transfer function is evaluated branch-free (tm_device_math.h bt709_eotf2_clamped).
// Edge rules as k_ingest_wave: an incomplete quad (odd last column / row) is not converted and reads as linear 0
// (cuda-colorspace/src/kernel.rs:64-65); downscale clamps (downscale.rs:22-30) become "take the in-range neighbour".
// grid (ceil(ceil(w/2) / 64), ceil(ceil(h/2) / (4 * rows_per_wave)), slots), block 256 = four independent waves that share one
// staging of the table; rows_per_wave even, <= 128.
// ------------------------------------------------------------------------------------------------
using ::tm_swap1; // (the one-float form: tm_platform.h)
__device__ __forceinline__ tmdev::tm_f2 tm_swap1(tmdev::tm_f2 v) { return tmdev::f2_make(tm_swap1(v.x), tm_swap1(v.y)); }

// ds4 on {ref, dis} pairs with wave-uniform-per-lane flags (level 2 only; level 1 never clamps: an incomplete quad is all zeros)
__device__ __forceinline__ tmdev::tm_f2 ds4_sides(tmdev::tm_f2 v00, tmdev::tm_f2 v01, tmdev::tm_f2 v10, tmdev::tm_f2 v11, bool okx, bool oky)
{
    using namespace tmdev;
    const tm_f2 b = okx ? v01 : v00;
    const tm_f2 c = oky ? v10 : v00;
    const tm_f2 d = okx ? (oky ? v11 : v01) : (oky ? v10 : v00);
    tm_f2 sum = f2_splat(0.0f) + v00;
    sum = sum + b; sum = sum + c; sum = sum + d;
    return sum * f2_splat(0.25f);
}

// the part of TmGeom this kernel reads (levels 0, 1, 2): a small kernarg keeps the scalar registers for the loop
struct TmIngestGeom {
    int w, h, w1, h1, w2, h2;
    int pitch0, pitch1, pitch2;
    unsigned long long plane0, plane1, plane2, off1, pyr;
    // levels 2..5 (the FOLD epilogue): sizes, pitches, plane sizes and offsets inside a pyramid
    int wu[4], hu[4], pitchu[4];
    unsigned long long planeu[4], offu[4];
};
__host__ __device__ inline TmIngestGeom tm_ingest_geom(const TmGeom &g)
{
    TmIngestGeom o{g.s[0].w, g.s[0].h, g.s[1].w, g.s[1].h, g.s[2].w, g.s[2].h, g.s[0].pitch, g.s[1].pitch, g.s[2].pitch,
                   g.s[0].plane, g.s[1].plane, g.s[2].plane, g.s[1].off, g.pyr, {}, {}, {}, {}, {}};
    for (int k = 0; k < 4; ++k) { o.wu[k] = g.s[2 + k].w; o.hu[k] = g.s[2 + k].h; o.pitchu[k] = g.s[2 + k].pitch; o.planeu[k] = g.s[2 + k].plane; o.offu[k] = g.s[2 + k].off; }
    return o;
}
// FOLD (round 6): levels 2..5 of the pyramid are finished by the workgroup that produced their level-2 linear pixels -- the four waves of
// a workgroup cover 128 x 16 rows_per_wave pixels, a whole number of level-5 pixels when rows_per_wave is 4 or 8 -- out of an LDS tile
// instead of the LIN2 arena and a second kernel (k_ingest_upper_rd: 3 MB written and read per 1080p pair, a launch, and a kernel of 2 040
// small workgroups at its own latency floor).  Same operations on the same values: same bits.
#define TM_FOLD_ROWS2 16 /* level-2 rows of the tile at rows_per_wave = 8 */

// QUANT: the launch wants the integer SSE (PSNR) and / or the u8 planes (SSIM, MS-SSIM); the SSIMULACRA2-only instantiation carries
// none of that code
// level-0 rows of the pyramid leave non-temporal: nothing reads them before the next kernel, and 4.2 GB per 64 1080p pairs only
// pass through the caches (measured on one engine, 1080p: ingest stage 1.19 -> 1.12 ms, the column pass that reads them
// unchanged; 4K: no difference)
#define TM_ROWS_STORE(v, p) __builtin_nontemporal_store(v, p)
// the SSIMULACRA2-only instantiation at five waves per SIMD (<= 96 VGPRs, five dwords of scratch) against four (107 VGPRs, none):
// ingest stage 1.31 vs 1.34 ms per 64 1080p pairs, 1.78 vs 1.97 ms per 24 4K pairs (tools/ingest_ab.py --distinct 32); the
// instantiation with the PSNR / u8-plane code needs 121 VGPRs and stays at four (held to 96 it spills: fused ingest 1.46 -> 2.10 ms)
#ifndef TM_ROWS_WAVES
#define TM_ROWS_WAVES 5
#endif
template <int KIND, bool QUANT, bool FOLD = false>
__global__ void __launch_bounds__(256) TM_WAVES_PER_SIMD(QUANT ? 4 : TM_ROWS_WAVES) k_ingest_rows(TmIngestGeom g, const TmFrameDesc *__restrict__ desc, const float *__restrict__ coef,
                                                    const double *__restrict__ gtab, float *__restrict__ XYB, float *__restrict__ LIN2,
                                                    unsigned long long *__restrict__ SSE, int want_sse, unsigned char *__restrict__ QU8,
                                                    unsigned long long qplane, int qpitch, int rows_per_wave)
{
    using namespace tmdev;
    static_assert(KIND == TM_KIND_NV12 || KIND == TM_KIND_P016 || KIND == TM_KIND_I420_8 || KIND == TM_KIND_I420_16 || KIND == TM_KIND_I420_P10, "4:2:0 kinds only");
    constexpr bool PLANAR = KIND == TM_KIND_I420_8 || KIND == TM_KIND_I420_16;
    constexpr bool P10 = KIND == TM_KIND_I420_P10;
    constexpr bool YUV8 = KIND == TM_KIND_NV12 || KIND == TM_KIND_I420_8;
    constexpr int BITS = YUV8 ? 8 : 16;
    using T = typename std::conditional<YUV8, unsigned char, unsigned short>::type;
    __shared__ __attribute__((aligned(32))) double et64[TM_EOTF64_STRIDE * TM_EOTF64_SEGS]; // the transfer-function table (tm_device_math.h bt709_power2), 16.4 KB per four waves
    // FOLD: {ref, dis} linear pixels of level 2 (32 columns x up to 16 rows per workgroup: 12.7 KB, filled while the waves walk); those of
    // levels 3 and 4 (3.3 + 0.9 KB) take the table's place once every wave is done with it -- 29.1 KB per workgroup: five per CU
    __shared__ tm_f2 l2t[FOLD ? 3 * TM_FOLD_ROWS2 * 33 : 1];
    tm_f2 *const l3t = (tm_f2 *)et64, *const l4t = l3t + 3 * (TM_FOLD_ROWS2 / 2) * 17;
    static_assert((3 * (TM_FOLD_ROWS2 / 2) * 17 + 3 * (TM_FOLD_ROWS2 / 4) * 9) * sizeof(tm_f2) <= TM_EOTF64_STRIDE * TM_EOTF64_SEGS * sizeof(double), "fold tiles fit the table");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // four independent waves share one staging of the table
    const int slot = blockIdx.z;
    const int w = g.w, h = g.h;
    const int qx = blockIdx.x * 64 + lane, X0 = 2 * qx;
    const int qy_begin = (blockIdx.y * 4 + wave) * rows_per_wave;
    const int qy_end = min(qy_begin + rows_per_wave, g.h1); // h1 == ceil(h / 2): quad rows, the incomplete last one included
    for (int i = threadIdx.x; i < TM_EOTF64_STRIDE * TM_EOTF64_SEGS; i += 256) { // {c0, c1, c2, c3}[k] -> {c0, c1}[k] | {c2, c3}[k] (bt709_power2)
        const int k = i >> 2, j = i & 3;
        et64[(j >> 1) * 2 * TM_EOTF64_SEGS + 2 * k + (j & 1)] = gtab[TM_TAB_EOTF64 + i];
    }
    const TmFrameDesc dd0 = desc[slot * 2], dd1 = desc[slot * 2 + 1];
    const float *kr = coef + (dd0.matrix * 2 + (BITS == 16 ? 1 : 0)) * 5, *kd = coef + (dd1.matrix * 2 + (BITS == 16 ? 1 : 0)) * 5;
    const tm_f2 k0 = f2_make(kr[0], kd[0]), k1 = f2_make(kr[1], kd[1]), k2 = f2_make(kr[2], kd[2]), k3 = f2_make(kr[3], kd[3]), k4 = f2_make(kr[4], kd[4]);
    const float neutral = (float)(1 << (BITS - 1)), ymin = (float)(16u << (BITS - 8));
    const bool colq = X0 + 1 < w; // this lane's quads are complete along x
    unsigned prn0[3] = {0, 0, 0}, prn1[3] = {0, 0, 0};
    // this lane's byte offsets inside a luma (CbCr) / a planar chroma row; P10: the wave's 128 luma samples are one run of the packed row, its
    // 64 chroma samples half a run, each behind a wave-uniform shift (tm_geom.h)
    const unsigned xo_y = P10 ? p10_word_offset((unsigned)(2 * qx)) : (unsigned)(2 * qx) * (unsigned)sizeof(T);
    const unsigned xo_c = P10 ? p10_word_offset((unsigned)qx) : (unsigned)qx * (unsigned)sizeof(T);
    const unsigned sy10 = __builtin_amdgcn_readfirstlane(p10_shift(blockIdx.x * 128u)), sc10 = __builtin_amdgcn_readfirstlane(p10_shift(blockIdx.x * 64u));
    const auto load_row = [&](const TmFrameDesc &d, int qyl, unsigned (&raw)[3]) {
        if (P10) yuv_row_load_p10(d, xo_y, xo_c, sy10, sc10, qyl, raw);
        else yuv_row_load_pairs<T, PLANAR>(d, xo_y, xo_c, qyl, raw);
    };
    if (colq && 2 * qy_begin + 1 < h) { load_row(dd0, qy_begin, prn0); load_row(dd1, qy_begin, prn1); }
    // (taken before the loop for the same reason as inside it: a load still pending at the loop header would make the compiler wait
    // for everything outstanding -- the previous row's stores included -- at the top of every iteration)
#pragma unroll
    for (int i = 0; i < 3; ++i) { TM_KEEP_IN_VGPR(prn0[i]); TM_KEEP_IN_VGPR(prn1[i]); }
    TM_LDS_BARRIER(); // the table is in place (the only barrier: from here on the waves never meet again)
    float *xi = XYB ? XYB + (size_t)slot * 2 * g.pyr : nullptr; // the slot's interleaved pyramid
    unsigned sse3[3] = {0, 0, 0};
    tm_f2 up[3] = {f2_splat(0.0f), f2_splat(0.0f), f2_splat(0.0f)}; // level-1 linear pixel of the quad row above (even rows wait here)
    const unsigned lane_b0 = (unsigned)X0 * 8u, lane_b1 = (unsigned)qx * 8u; // byte offsets of this lane's {ref, dis} pairs in a level-0 / level-1 row
#pragma unroll 1
    for (int qy = qy_begin; qy < qy_end; ++qy) {
        const int Y0 = 2 * qy;
        const bool quad_ok = colq && Y0 + 1 < h; // incomplete quads are not converted (cuda-colorspace/src/kernel.rs:64-65)
        unsigned pr0[3] = {prn0[0], prn0[1], prn0[2]}, pr1[3] = {prn1[0], prn1[1], prn1[2]};
        if (qy + 1 < qy_end && colq && Y0 + 3 < h) { // the next row's samples, requested before this row's arithmetic
            load_row(dd0, qy + 1, prn0); load_row(dd1, qy + 1, prn1);
        }
        // ---- biplanar.rs:8-70 on {ref, dis} pairs
        tm_f2 pr[4], pg[4], pb[4];
        {
            unsigned ra[6], rb[6];
            yuv_quad_unpack<BITS>(pr0, ra);
            yuv_quad_unpack<BITS>(pr1, rb);
            const tm_f2 cb = f2_make((float)ra[4], (float)rb[4]) - f2_splat(neutral), cr = f2_make((float)ra[5], (float)rb[5]) - f2_splat(neutral);
            const tm_f2 r_ = k1 * cr;
            const tm_f2 g_ = f2_fma(k3, cb, k4 * cr);
            const tm_f2 b_ = k2 * cb;
            tm_f2 vr[4], vg[4], vb[4];
            float vmin = 1.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const tm_f2 luma = (f2_make(fmaxf((float)ra[q], ymin), fmaxf((float)rb[q], ymin)) - f2_splat(ymin)) * k0;
                vr[q] = luma + r_; vg[q] = luma + g_; vb[q] = luma + b_;
                vmin = fminf(fminf(vmin, fminf(vr[q].x, vr[q].y)), fminf(fminf(vg[q].x, vg[q].y), fminf(vb[q].x, vb[q].y)));
                pr[q] = bt709_power2(vr[q], et64);
                pg[q] = bt709_power2(vg[q], et64);
                pb[q] = bt709_power2(vb[q], et64);
            }
            // the linear branch (v < 0.0812: luma codes below ~35) is rare in pictures: a wave evaluates it only when one of
            // its 24 x 64 arguments needs it (same bits either way)
            if (TM_WAVE_ANY(!(vmin >= 0.08124285829863521110029445797874f))) {
                TM_NO_IF_CONVERSION();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pr[q] = bt709_eotf_linear_fix(vr[q], pr[q]);
                    pg[q] = bt709_eotf_linear_fix(vg[q], pg[q]);
                    pb[q] = bt709_eotf_linear_fix(vb[q], pb[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { pr[q] = clamp01_2(pr[q]); pg[q] = clamp01_2(pg[q]); pb[q] = clamp01_2(pb[q]); }
            if (TM_WAVE_ANY(!quad_ok)) { // only the waves on the right / bottom edge of an odd-sized frame
                TM_NO_IF_CONVERSION();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pr[q] = quad_ok ? pr[q] : f2_splat(0.0f); pg[q] = quad_ok ? pg[q] : f2_splat(0.0f); pb[q] = quad_ok ? pb[q] : f2_splat(0.0f);
                }
            }
        }
        if (QUANT) { // (see below: with the u8-plane stores ahead, the next row's samples are taken before them)
#pragma unroll
            for (int i = 0; i < 3; ++i) { TM_KEEP_IN_VGPR(prn0[i]); TM_KEEP_IN_VGPR(prn1[i]); }
        }
        if (QUANT) { // sample_conv.rs:6-35 quantisation; out-of-image samples are 0 on both sides
            const tm_f2 *pc[3] = {pr, pg, pb};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float ssum = 0.0f; // <= 4 * 255^2: exact in f32
                unsigned qa[4], qb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const tm_f2 v = pc[c][q] * f2_splat(255.0f);
                    const float fa = rintf(v.x), fb = rintf(v.y), dlt = fa - fb;
                    ssum += dlt * dlt;
                    qa[q] = (unsigned)(int)fa; qb[q] = (unsigned)(int)fb;
                }
                sse3[c] += (unsigned)(int)ssum;
                if (QU8 != nullptr && X0 < w) { // u8-quantised planes for SSIM / MS-SSIM: two pixels = one 16-bit store per row
#pragma unroll
                    for (int iy = 0; iy < 2; ++iy)
                        if (Y0 + iy < h) {
                            const size_t o = tm_mul24((unsigned)(Y0 + iy), (unsigned)qpitch) + (unsigned)X0;
                            *(unsigned short *)(QU8 + ((size_t)(slot * 2 + 0) * 3 + c) * qplane + o) = (unsigned short)(qa[2 * iy] | (qa[2 * iy + 1] << 8));
                            *(unsigned short *)(QU8 + ((size_t)(slot * 2 + 1) * 3 + c) * qplane + o) = (unsigned short)(qb[2 * iy] | (qb[2 * iy + 1] << 8));
                        }
                }
            }
        }
        if (xi == nullptr) continue; // PSNR / SSIM only: no pyramid (wave-uniform)
        // ---- level 1: downscale.rs:22-30 (never clamps here: a quad is complete or all zeros), then xyb.rs:42-79 for the 4 + 1 pixels
        tm_f2 lr[5], lg[5], lb[5], xa[5], xb[5], xc[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lr[q] = pr[q]; lg[q] = pg[q]; lb[q] = pb[q]; }
        lr[4] = (((f2_splat(0.0f) + pr[0]) + pr[1]) + pr[2] + pr[3]) * f2_splat(0.25f);
        lg[4] = (((f2_splat(0.0f) + pg[0]) + pg[1]) + pg[2] + pg[3]) * f2_splat(0.25f);
        lb[4] = (((f2_splat(0.0f) + pb[0]) + pb[1]) + pb[2] + pb[3]) * f2_splat(0.25f);
        linear_to_xyb_sides<5>(lr, lg, lb, xa, xb, xc);
        // the next row's samples (requested at the top of this iteration) are taken HERE, before this row's fifteen stores are
        // issued: gfx950 counts stores in vmcnt in order, so waiting for those loads at the loop's latch would wait for the stores too
        if (!QUANT) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { TM_KEEP_IN_VGPR(prn0[i]); TM_KEEP_IN_VGPR(prn1[i]); }
        }
        {
            const tm_f2 *xv[3] = {xa, xb, xc};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
                    if (X0 < w && Y0 + iy < h) { // X0 is even and the pitch a multiple of 64 floats: the pair of pixels stays inside the row
                        TM_GLOBAL_AS char *rowp = (TM_GLOBAL_AS char *)tm_uniform_ptr(xi + 2 * (c * g.plane0) + 2 * (size_t)(Y0 + iy) * g.pitch0);
                        TM_ROWS_STORE(tm_make_f4(xv[c][2 * iy].x, xv[c][2 * iy].y, xv[c][2 * iy + 1].x, xv[c][2 * iy + 1].y), (TM_GLOBAL_AS tm_f4 *)(rowp + lane_b0));
                    }
                if (qx < g.w1) { // qy < h1 by the loop bound
                    TM_GLOBAL_AS char *rowp = (TM_GLOBAL_AS char *)tm_uniform_ptr(xi + 2 * (g.off1 + c * g.plane1) + 2 * (size_t)qy * g.pitch1);
                    *(TM_GLOBAL_AS tm_g2 *)(rowp + lane_b1) = tm_g2{xv[c][4].x, xv[c][4].y};
                }
            }
        }
        // ---- level-2 LINEAR pixel of each 2 x 2 group of level-1 pixels (levels 2..5 are finished by k_ingest_upper_rd): rows pair
        // up across two iterations -- qy_begin is even --, columns across lane ^ 1
        const tm_f2 l1[3] = {lr[4], lg[4], lb[4]};
        const bool lower = (qy & 1) != 0, last_alone = !lower && qy + 1 == g.h1; // a last even row has no partner: oky = false
        if (lower || last_alone) {
            const int XL = qx >> 1, YL = qy >> 1;
            const bool ok2x = 2 * XL + 1 < g.w1, ok2y = lower;
            const bool st = !(lane & 1) && XL < g.w2; // YL < h2: row 2 YL exists
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const tm_f2 v00 = lower ? up[c] : l1[c], v10 = l1[c];
                const tm_f2 v = ds4_sides(v00, tm_swap1(v00), v10, tm_swap1(v10), ok2x, ok2y);
                if (FOLD) {
                    if (st) l2t[(c * TM_FOLD_ROWS2 + (YL - (int)blockIdx.y * 2 * rows_per_wave)) * 33 + (lane >> 1)] = v;
                } else if (st) {
                    const size_t o = (size_t)c * g.plane2 + tm_mul24((unsigned)YL, (unsigned)g.pitch2) + (unsigned)XL;
                    LIN2[(size_t)(slot * 2 + 0) * 3 * g.plane2 + o] = v.x;
                    LIN2[(size_t)(slot * 2 + 1) * 3 * g.plane2 + o] = v.y;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) up[c] = l1[c];
        }
    }
    if (QUANT && want_sse) {
        if (tm_wave_sum_u32x3(sse3)) {
            const unsigned bin = (blockIdx.x + (blockIdx.y * 4 + wave) * 29) % TM_SSE_BINS;
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicAdd(&SSE[((size_t)slot * TM_SSE_BINS + bin) * 3 + c], (unsigned long long)sse3[c]);
        }
    }
    if (!FOLD || xi == nullptr) return; // (wave- and workgroup-uniform)
    // ---- levels 2..5 of this workgroup's tile: 32 x R2 pixels of level 2 (R2 = 2 rows_per_wave = 8 or 16), what k_ingest_upper_rd does for
    // the tile kernel (downscale.rs:5-35, xyb.rs:42-79), on {ref, dis} pairs.  Pixels beyond the image read as 0 and are never stored.
    TM_LDS_BARRIER(); // the four waves' level-2 pixels are in the tile
    const int tid = threadIdx.x;
    const int R2 = 2 * rows_per_wave, x2_0 = blockIdx.x * 32, y2_0 = blockIdx.y * R2;
    if (tid < 8 * R2) { // one lane per 2 x 2 quad of level 2: its four pixels and their level-3 parent
        const int qx2 = tid & 15, qy2 = tid >> 4, X2 = x2_0 + 2 * qx2, Y2 = y2_0 + 2 * qy2;
        tm_f2 lr[5], lg[5], lb[5], xa[5], xb[5], xc[5];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = X2 + (k & 1) < g.wu[0] && Y2 + (k >> 1) < g.hu[0];
            const int o = (2 * qy2 + (k >> 1)) * 33 + 2 * qx2 + (k & 1);
            const tm_f2 a = l2t[o], b = l2t[TM_FOLD_ROWS2 * 33 + o], c = l2t[2 * TM_FOLD_ROWS2 * 33 + o];
            lr[k] = in ? a : f2_splat(0.0f); lg[k] = in ? b : f2_splat(0.0f); lb[k] = in ? c : f2_splat(0.0f);
        }
        const bool okx = X2 + 1 < g.wu[0], oky = Y2 + 1 < g.hu[0];
        lr[4] = ds4_sides(lr[0], lr[1], lr[2], lr[3], okx, oky);
        lg[4] = ds4_sides(lg[0], lg[1], lg[2], lg[3], okx, oky);
        lb[4] = ds4_sides(lb[0], lb[1], lb[2], lb[3], okx, oky);
        l3t[qy2 * 17 + qx2] = lr[4]; l3t[(TM_FOLD_ROWS2 / 2) * 17 + qy2 * 17 + qx2] = lg[4]; l3t[2 * (TM_FOLD_ROWS2 / 2) * 17 + qy2 * 17 + qx2] = lb[4];
        linear_to_xyb_sides<5>(lr, lg, lb, xa, xb, xc);
        const tm_f2 *xv[3] = {xa, xb, xc};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int iy = 0; iy < 2; ++iy)
                if (X2 < g.wu[0] && Y2 + iy < g.hu[0]) // X2 is even and the pitch a multiple of 64 floats: the pair of pixels stays inside the row
                    *(tm_f4 *)(xi + 2 * (g.offu[0] + c * g.planeu[0] + (size_t)(Y2 + iy) * g.pitchu[0] + X2)) =
                        tm_make_f4(xv[c][2 * iy].x, xv[c][2 * iy].y, xv[c][2 * iy + 1].x, xv[c][2 * iy + 1].y);
            if (X2 / 2 < g.wu[1] && Y2 / 2 < g.hu[1])
                *(tm_g2 *)(xi + 2 * (g.offu[1] + c * g.planeu[1] + (size_t)(Y2 / 2) * g.pitchu[1] + X2 / 2)) = tm_g2{xv[c][4].x, xv[c][4].y};
        }
    }
    TM_LDS_BARRIER();
    if (tid < 2 * R2) { // level 4: 8 x R2 / 4 pixels
        const int ox = tid & 7, oy = tid >> 3, XL = (x2_0 >> 2) + ox, YL = (y2_0 >> 2) + oy;
        const bool okx = 2 * XL + 1 < g.wu[1], oky = 2 * YL + 1 < g.hu[1];
        tm_f2 v[3][1], xa[1], xb[1], xc[1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const tm_f2 *t = l3t + c * (TM_FOLD_ROWS2 / 2) * 17 + (2 * oy) * 17 + 2 * ox;
            v[c][0] = ds4_sides(t[0], t[1], t[17], t[18], okx, oky);
            l4t[c * (TM_FOLD_ROWS2 / 4) * 9 + oy * 9 + ox] = v[c][0];
        }
        if (XL < g.wu[2] && YL < g.hu[2]) {
            linear_to_xyb_sides<1>(v[0], v[1], v[2], xa, xb, xc);
            const size_t o = g.offu[2] + (size_t)YL * g.pitchu[2] + XL;
            *(tm_g2 *)(xi + 2 * o) = tm_g2{xa[0].x, xa[0].y};
            *(tm_g2 *)(xi + 2 * (o + g.planeu[2])) = tm_g2{xb[0].x, xb[0].y};
            *(tm_g2 *)(xi + 2 * (o + 2 * g.planeu[2])) = tm_g2{xc[0].x, xc[0].y};
        }
    }
    TM_LDS_BARRIER();
    if (tid < R2 / 2) { // level 5: 4 x R2 / 8 pixels
        const int ox = tid & 3, oy = tid >> 2, XL = (x2_0 >> 3) + ox, YL = (y2_0 >> 3) + oy;
        const bool okx = 2 * XL + 1 < g.wu[2], oky = 2 * YL + 1 < g.hu[2];
        tm_f2 v[3][1], xa[1], xb[1], xc[1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const tm_f2 *t = l4t + c * (TM_FOLD_ROWS2 / 4) * 9 + (2 * oy) * 9 + 2 * ox;
            v[c][0] = ds4_sides(t[0], t[1], t[9], t[10], okx, oky);
        }
        if (XL < g.wu[3] && YL < g.hu[3]) {
            linear_to_xyb_sides<1>(v[0], v[1], v[2], xa, xb, xc);
            const size_t o = g.offu[3] + (size_t)YL * g.pitchu[3] + XL;
            *(tm_g2 *)(xi + 2 * o) = tm_g2{xa[0].x, xa[0].y};
            *(tm_g2 *)(xi + 2 * (o + g.planeu[3])) = tm_g2{xb[0].x, xb[0].y};
            *(tm_g2 *)(xi + 2 * (o + 2 * g.planeu[3])) = tm_g2{xc[0].x, xc[0].y};
        }
    }
}

// Levels 2..5 of the pyramid from the level-2 linear RGB that k_ingest_wave leaves in LIN2 (1/16 of the pixels): XYB of
// level 2, then 2x2 box downscales (downscale.rs:5-35) and XYB for levels 3, 4, 5.  Workgroup = 32x32 tile of level 2
// (= 128x128 px of level 0, so every parent stays in the tile), lane = 2x2 quad, both sides in one workgroup: side 0's XYB
// values (five per lane for levels 2 and 3, one each for the lanes that own a level-4 / level-5 pixel) wait in registers
// and side 1 stores whole {ref, dis} pairs -- a float4 per level-2 row (two pixels x two sides: 16 lanes = 256 contiguous
// bytes).  grid (ceil(w2/32), ceil(h2/32), slots), block 256.
__global__ void __launch_bounds__(256) k_ingest_upper_rd(TmGeom g, const float *__restrict__ LIN2, float *__restrict__ XYB)
{
    __shared__ float lin3[3][16][17]; // level-3 linear RGB of this tile
    __shared__ float lin4[3][8][9];
    const int tid = threadIdx.x, qx = tid & 15, qy = tid >> 4;
    const int slot = blockIdx.z;
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;
    const TmScaleGeom s2 = g.s[2], s3 = g.s[3], s4 = g.s[4], s5 = g.s[5];
    float *xi = XYB + (size_t)slot * 2 * g.pyr;
    const int X0 = tx0 + 2 * qx, Y0 = ty0 + 2 * qy;
    float k23[3][5], k4[3] = {0, 0, 0}, k5[3] = {0, 0, 0}; // side 0
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) k23[c][k] = 0.0f;
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
        const float *l2 = LIN2 + (size_t)(slot * 2 + side) * 3 * s2.plane;
        float lr[5], lg[5], lb[5], xa[5], xb[5], xc[5];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = X0 + (k & 1), y = Y0 + (k >> 1);
            const bool in = x < s2.w && y < s2.h;
            const size_t o = (size_t)(in ? y : 0) * s2.pitch + (in ? x : 0);
            const float a = l2[o], b = l2[s2.plane + o], c = l2[2 * s2.plane + o];
            lr[k] = in ? a : 0.0f; lg[k] = in ? b : 0.0f; lb[k] = in ? c : 0.0f;
        }
        const bool okx = X0 + 1 < s2.w, oky = Y0 + 1 < s2.h;
        lr[4] = ds4(lr[0], lr[1], lr[2], lr[3], okx, oky);
        lg[4] = ds4(lg[0], lg[1], lg[2], lg[3], okx, oky);
        lb[4] = ds4(lb[0], lb[1], lb[2], lb[3], okx, oky);
        lin3[0][qy][qx] = lr[4]; lin3[1][qy][qx] = lg[4]; lin3[2][qy][qx] = lb[4];
        tmdev::linear_to_xyb_n<5>(lr, lg, lb, xa, xb, xc);
        const float *xv[3] = {xa, xb, xc};
        if (side == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 5; ++k) k23[c][k] = xv[c][k];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
                    if (X0 < s2.w && Y0 + iy < s2.h) // X0 is even and the pitch a multiple of 64 floats: the pair of pixels stays inside the row
                        *(float4 *)(xi + 2 * (s2.off + c * s2.plane + (size_t)(Y0 + iy) * s2.pitch + X0)) =
                            make_float4(k23[c][2 * iy], xv[c][2 * iy], k23[c][2 * iy + 1], xv[c][2 * iy + 1]);
                if (X0 / 2 < s3.w && Y0 / 2 < s3.h)
                    *(float2 *)(xi + 2 * (s3.off + c * s3.plane + (size_t)(Y0 / 2) * s3.pitch + X0 / 2)) = make_float2(k23[c][4], xv[c][4]);
            }
        }
        TM_LDS_BARRIER();
        if (tid < 64) { // level 4: 8x8 per tile
            const int ox = tid & 7, oy = tid >> 3;
            const int XL = (tx0 >> 2) + ox, YL = (ty0 >> 2) + oy;
            const bool ok4x = 2 * XL + 1 < s3.w, ok4y = 2 * YL + 1 < s3.h;
            float v[3], a, b, c;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                v[ch] = ds4(lin3[ch][2 * oy][2 * ox], lin3[ch][2 * oy][2 * ox + 1], lin3[ch][2 * oy + 1][2 * ox], lin3[ch][2 * oy + 1][2 * ox + 1], ok4x, ok4y);
                lin4[ch][oy][ox] = v[ch];
            }
            if (XL < s4.w && YL < s4.h) {
                tmdev::linear_to_xyb(v[0], v[1], v[2], a, b, c);
                if (side == 0) { k4[0] = a; k4[1] = b; k4[2] = c; }
                else {
                    const size_t o = s4.off + (size_t)YL * s4.pitch + XL;
                    *(float2 *)(xi + 2 * o) = make_float2(k4[0], a);
                    *(float2 *)(xi + 2 * (o + s4.plane)) = make_float2(k4[1], b);
                    *(float2 *)(xi + 2 * (o + 2 * s4.plane)) = make_float2(k4[2], c);
                }
            }
        }
        TM_LDS_BARRIER();
        if (tid < 16) { // level 5: 4x4 per tile
            const int ox = tid & 3, oy = tid >> 2;
            const int XL = (tx0 >> 3) + ox, YL = (ty0 >> 3) + oy;
            const bool ok5x = 2 * XL + 1 < s4.w, ok5y = 2 * YL + 1 < s4.h;
            float v[3], a, b, c;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                v[ch] = ds4(lin4[ch][2 * oy][2 * ox], lin4[ch][2 * oy][2 * ox + 1], lin4[ch][2 * oy + 1][2 * ox], lin4[ch][2 * oy + 1][2 * ox + 1], ok5x, ok5y);
            if (XL < s5.w && YL < s5.h) {
                tmdev::linear_to_xyb(v[0], v[1], v[2], a, b, c);
                if (side == 0) { k5[0] = a; k5[1] = b; k5[2] = c; }
                else {
                    const size_t o = s5.off + (size_t)YL * s5.pitch + XL;
                    *(float2 *)(xi + 2 * o) = make_float2(k5[0], a);
                    *(float2 *)(xi + 2 * (o + s5.plane)) = make_float2(k5[1], b);
                    *(float2 *)(xi + 2 * (o + 2 * s5.plane)) = make_float2(k5[2], c);
                }
            }
        }
        TM_LDS_BARRIER(); // lin3 / lin4 are reused by the next side
    }
}

template <int R> struct BlurVTile {
    static constexpr int S = R == 32 ? 65 : (R == 16 ? 66 : 68);
    static constexpr int LPC = R / 4;   // lanes per column in the read-back
    static constexpr int CPI = 64 / LPC; // columns per store instruction
};

__device__ __forceinline__ float ld_row_u(const float *__restrict__ plane, unsigned xb, int row, int nrows, int pitch)
{
    // xb = this lane's BYTE offset inside the row (a zero-extended 32-bit VGPR offset is what the
    // scalar-base form of global_load takes)
    const int rc = row < nrows ? row : nrows - 1;
    TM_GLOBAL_AS const char *rowp = (TM_GLOBAL_AS const char *)tm_uniform_ptr(plane + (size_t)rc * pitch);
    const float v = *(TM_GLOBAL_AS const float *)(rowp + xb);
    return row < nrows ? v : 0.0f;
}

__device__ __forceinline__ void ld_row_u2(const float *__restrict__ plane, unsigned xb, int row, int nrows, int pitch, float &a, float &b)
{
    // the {ref, dis} pair of the interleaved pyramid: one 8-byte load per lane
    const int rc = row < nrows ? row : nrows - 1;
    TM_GLOBAL_AS const char *rowp = (TM_GLOBAL_AS const char *)tm_uniform_ptr(plane + (size_t)rc * pitch);
    const tm_g2 v = *(TM_GLOBAL_AS const tm_g2 *)(rowp + xb);
    a = row < nrows ? v.x : 0.0f;
    b = row < nrows ? v.y : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Column pass ("pass 1"), tuned: blur_plane_pass_fused down the columns of the five planes ref^2, dis^2, ref*dis, ref, dis
// (ssimulacra2-cuda-kernel/src/blur.rs:34-137; which planes: ssimulacra2-cuda/src/lib.rs:299-335), products formed in
// registers (a rounded f32 multiply each, exactly what nppiMul stores), output transposed (the reference's nppiTranspose,
// lib.rs:342-361) as whole 128-B lines.  Exact sequential recurrence per column: step t reads row t and row t-10 and emits
// row t-4.  One workgroup = 5 wavefronts = the 5 planes of one 64-column block; each wave runs ONE recurrence per lane
// (6 state registers), keeps a W-row register window of its input (the reference's 11-deep ring + a W-10 row load prefetch;
// static slot indices via unroll) and owns one [32][65] LDS tile: every 32 steps the wave reads its tile back transposed
// (row stride 65 -> conflict-free both ways) and stores, per column, 32 contiguous floats, non-temporal.
// The input is the ref/dis-interleaved pyramid (a pixel = the 8-byte pair {ref, dis}, rows of 2 * pitch floats):
//   PAIR = false ("pair lanes"): the wave covers 32 columns, lane = (column, side): one coalesced 256-B load per row feeds
//          the ref AND the dis recurrence, and the flush sends the odd tile columns to the next output plane (side_delta
//          floats further).  square = true: the input is squared (planes 0, 1), false: plain (planes 3, 4).
//   PAIR = true: the product wave, 64 columns, one 8-byte {ref, dis} load per lane -> plane 2.
// Flush stores are unconditional (planes are padded) so the loop body is one basic block and every load gets an exact
// s_waitcnt vmcnt(N) (gfx950 counts stores in vmcnt, in order).
// ------------------------------------------------------------------------------------------------
template <int R, int W, bool PAIR>
__device__ __forceinline__ void blur_v_role(float *__restrict__ tile, const float *__restrict__ in, unsigned x, float *__restrict__ dst,
                                            int h, int pitch, int pitch_t, bool square, unsigned side_delta)
{
    // in: wave-uniform base of the interleaved input plane; x: this lane's BYTE offset inside a row; dst: wave-uniform pointer
    // to transposed row x0 of the (first) output plane
    using TT = BlurVTile<R>;
    static_assert(TT::CPI % 2 == 0, "pair lanes: two tile columns per image column");
    constexpr int P = W - 10;
    constexpr int U = W > R ? W : R;
    constexpr int XS = PAIR ? 1 : 2; // tile columns per image column
    const int lane = threadIdx.x & 63;
    const int xl = lane / TT::LPC, yq = lane % TT::LPC;
    // per-lane BYTE part of every flush address
    const unsigned voff = PAIR ? (unsigned)(xl * pitch_t + 4 * yq) * 4u
                               : (unsigned)((xl >> 1) * pitch_t + 4 * yq) * 4u + (unsigned)(xl & 1) * side_delta * 4u;
    float wa[W], wb[PAIR ? W : 1];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        if (PAIR) {
            if (j < P) ld_row_u2(in, x, j, h, pitch, wa[j], wb[j]);
            else wa[j] = wb[j] = 0.0f;
        } else wa[j] = j < P ? ld_row_u(in, x, j, h, pitch) : 0.0f;
    }
    tmdev::Iir f = {0, 0, 0, 0, 0, 0};
    const bool product = PAIR || square;
#pragma unroll
    for (int t = 0; t < 4; ++t) { // no output row yet
        const float a = wa[t], aold = wa[(t + P) % W];
        const float b = PAIR ? wb[t] : a, bold = PAIR ? wb[(t + P) % W] : aold;
        if (PAIR) ld_row_u2(in, x, t + P, h, pitch, wa[(t + P) % W], wb[(t + P) % W]);
        else wa[(t + P) % W] = ld_row_u(in, x, t + P, h, pitch);
        (void)tmdev::iir_step(f, product ? aold * bold + a * b : aold + a);
    }
    const int T = (h + U - 1) / U * U + 4;
    for (int t0 = 4; t0 < T; t0 += U) {
        // Keep the five role-waves of the workgroup within one iteration of each other: they read the same
        // ref/dis rows (3 readers each), and only while they stay close do the 2nd and 3rd reader hit L1/L2
        // (rocprofv3 FETCH_SIZE showed ~2.6x the algorithmic read bytes without this).  A bare s_barrier: no
        // data is exchanged, so nothing has to be drained (no vmcnt(0) as __syncthreads would add).
        __builtin_amdgcn_s_barrier();
        // one step; LOADROW: how row t + P reaches the window
#define TM_BLUR_V_STEP(LOADROW)                                                                                                      \
        {                                                                                                                            \
            const int t = t0 + j;                                                                                                    \
            const float a = wa[(j + 4) % W], aold = wa[(j + 4 + P) % W];                                                             \
            const float b = PAIR ? wb[(j + 4) % W] : a, bold = PAIR ? wb[(j + 4 + P) % W] : aold;                                    \
            LOADROW;                                                                                                                 \
            const float o = tmdev::iir_step(f, product ? aold * bold + a * b : aold + a);                                            \
            tile[(j % R) * TT::S + lane] = o;                                                                                        \
            if (j % R == R - 1) {                                                                                                    \
                const int y0 = t - 4 - (R - 1);                                                                                      \
                __builtin_amdgcn_wave_barrier();                                                                                     \
                _Pragma("unroll") for (int i = 0; i < 64 / TT::CPI; ++i) {                                                           \
                    const int xc = i * TT::CPI + xl;                                                                                 \
                    const float *tp = tile + (4 * yq) * TT::S + xc;                                                                  \
                    TM_GLOBAL_AS char *ub = (TM_GLOBAL_AS char *)tm_uniform_ptr(dst + (size_t)(i * TT::CPI / XS) * pitch_t + y0);   \
                    __builtin_nontemporal_store(tm_make_f4(tp[0], tp[TT::S], tp[2 * TT::S], tp[3 * TT::S]), (TM_GLOBAL_AS tm_f4 *)(ub + voff)); \
                }                                                                                                                    \
                __builtin_amdgcn_wave_barrier();                                                                                     \
            }                                                                                                                        \
        }
        // (round 4: a second copy of the block without clamps and selects -- every row of an interior block exists -- with the row address
        // advanced by two scalar adds per step was 21 instead of 30 instructions per step and SLOWER: one 1080p pair 0.101 -> 0.117 ms,
        // one 4K pair 0.22 -> 0.27; the step is not bound by its instruction count: profiles/r04i_split10.log)
        {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (PAIR) {
                    TM_BLUR_V_STEP(ld_row_u2(in, x, t + P, h, pitch, wa[(j + 4 + P) % W], wb[(j + 4 + P) % W]))
                } else {
                    TM_BLUR_V_STEP(wa[(j + 4 + P) % W] = ld_row_u(in, x, t + P, h, pitch))
                }
            }
        }
#undef TM_BLUR_V_STEP
    }
}

// Job-table driven (tm_geom.h).  FULL job: waves 0, 1 = squares of columns 0..31 / 32..63 (-> planes 0 and 1), wave 2 = the
// product (-> plane 2), waves 3, 4 = plain values (-> planes 3 and 4).  EDGE job (only mu1, mu2 carry weight: scale 0 of the
// X and B channels, i.e. half of all pixels): waves 0..3 = plain values of four neighbouring 32-column blocks, wave 4 retires
// at once; it reads 2 and writes 2 planes instead of 2 + 5.  Every row load is a whole run of 256 or 512 bytes.
// The grid is slot-major (x = slot, y = block): workgroups are dispatched x-fastest, so the long jobs (scale 0) of ALL slots
// start first and the short scales fill the tail (longest-processing-time order).  grid (slots, jobs.vstart[n]), block 320.
__device__ __forceinline__ int tm_find_job(const int (&start)[TM_MAX_JOBS + 1], int b)
{
    int j = 0; // padding entries hold the total, so they never match
#pragma unroll
    for (int i = 1; i < TM_MAX_JOBS; ++i)
        if (b >= start[i]) j = i;
    return j;
}

// PROBE: a second instantiation of the same code for the placement probe of tm_engine_create, so that profilers list its
// launches (cold caches, zeros) apart from the batch launches
// (W = 32 -- 22 rows of loads in flight instead of 6 -- was measured in round 4 for launches that leave most of the chip idle: one 1080p pair
// 0.104 -> 0.101 ms, two pairs slower; large launches are bound by their write stream anyway, docs/LABBOOK.md 5.1: not kept)
// SOLO (launches of a pair or two): the five role-waves of a column block as five single-wave workgroups (grid z = role) -- on an idle
// chip every wave then has a SIMD to itself, while the five waves of one workgroup share the four SIMDs of one CU and the pair that
// shares a SIMD sets the pace of all five (they meet at the barrier): one 1080p pair 0.105 -> 0.094 ms, 2.94 k -> 3.05 k pairs/s (a lone wave
// still takes ~7 cycles per instruction: the gain is the pair that no longer shares).  No barrier, so the 2nd and 3rd reader of a ref / dis
// row miss the caches more often: from two pairs per launch on the five-wave workgroups win again (profiles/r04s_solo_col.log).
template <int R, int W, int PROBE = 0, bool SOLO = false>
__global__ void __launch_bounds__(SOLO ? 64 : 320, 4) k_blur_v_jobs(TmGeom g, TmJobs jobs, const float *__restrict__ XYB, float *__restrict__ V)
{
    using TT = BlurVTile<R>;
    __shared__ float tiles[(SOLO ? 1 : 5) * R * TT::S];
    // the fused kernel of the EDGE jobs runs beside this pass (k_blur_edge_fused, second stream): its waves are the oldest on their
    // SIMDs and would win every issue arbitration; this pass needs few issue slots but needs them promptly to keep HBM busy
    if (jobs.prio > 0) TM_SETPRIO(2);
    const int b = blockIdx.y, slot = blockIdx.x;
    const int j = tm_find_job(jobs.vstart, b);
    const int s = jobs.scale[j], c = jobs.chan[j], mode = jobs.mode[j];
    const TmScaleGeom sg = g.s[s];
    const int wave = SOLO ? (int)blockIdx.z : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform
    const int lane = threadIdx.x & 63;
    float *tile = tiles + (SOLO ? 0 : wave) * R * TT::S;
    const float *in = XYB + (size_t)slot * 2 * g.pyr + 2 * (sg.off + c * sg.plane);
    int blk = b - jobs.vstart[j], half, role;
    if (mode == TM_MODE_FULL) { role = wave == 2 ? 2 : (wave < 2 ? 0 : 3); half = wave == 2 ? 0 : (wave < 2 ? wave : wave - 3); }
    else {
        if (wave == 4) return; // a retired wave no longer counts at s_barrier
        role = 3; half = wave & 1; blk = blk * 2 + (wave >> 1);
        if (blk * 64 >= sg.w) return;
    }
    const int x0 = blk * 64;
    float *vdst = V + (size_t)(slot * 5 + role) * g.pyr_t + sg.off_t + c * sg.plane_t + (size_t)(x0 + 32 * half) * sg.pitch_t;
    if (role == 2) {
        const unsigned xc = (unsigned)min(x0 + lane, sg.w - 1) * 8u; // lanes past the right edge shadow the last column
        blur_v_role<R, W, true>(tile, in, xc, vdst, sg.h, 2 * sg.pitch, sg.pitch_t, true, 0u);
    } else {
        const unsigned xp = (unsigned)min(x0 + 32 * half + (lane >> 1), sg.w - 1) * 8u + (unsigned)(lane & 1) * 4u;
        blur_v_role<R, W, false>(tile, in, xp, vdst, sg.h, 2 * sg.pitch, sg.pitch_t, role == 0, (unsigned)g.pyr_t);
    }
}

// ------------------------------------------------------------------------------------------------
// Row pass, tuned ("x" = transposes ref / dis itself): same arithmetic as blur_h_job, one lane per image row walking x.
// The five (FULL) or two (EDGE) blurred planes come from the transposed V arena (lanes = consecutive y: coalesced 256-B
// reads) through WN-slot register windows: rows t-10 .. t+WN-11 are in registers or in flight.  The two edge-term inputs
// ref(x, y), dis(x, y) exist only in the normal orientation, in the interleaved pyramid: the wave fetches them in blocks of 16
// columns -- one 8-byte {ref, dis} load per lane = 4 rows x 16 pairs = four whole 128-B lines --, parks each 64 x 16 block in
// a double-buffered [64][17] LDS tile per side and reads it back one column per step, one row per lane (this replaces the
// reference's nppiTranspose of ref / dis, lib.rs:383-390).  Loads run D steps ahead of their LDS write and a whole block
// ahead of their use: element e = 16 * block + row group is requested at step u = e - 16 - D, written at u = e - 16,
// consumed during steps 16 * block .. + 15 (u = t - 4 = the column whose maps are evaluated at step t).  Everything stays
// inside the wave: LDS operations of one wave execute in order.  17.4 KB of LDS per wave.
// ------------------------------------------------------------------------------------------------
template <bool FULL, int WN, int D>
__device__ __forceinline__ void blur_h_job_x(float (*__restrict__ tile)[2][64][17], const float *__restrict__ rdn,
                                             const float *__restrict__ v0, const float *__restrict__ v1,
                                             const float *__restrict__ v2, const float *__restrict__ v3,
                                             const float *__restrict__ v4, int y0, int w, int h,
                                             int pitch, int pt, bool valid, double (&acc)[6])
{
    // rdn: interleaved plane of this channel ({ref, dis} of row y at 2 * y * pitch); v0..v4: transposed planes + this lane's row
    static_assert(WN % D == 0 && D <= 16, "queue depth");
    constexpr int P = WN - 10; // load distance of the blurred planes, in rows of the transposed arena
    constexpr int NF = FULL ? WN : 1;
    const int lane = threadIdx.x & 63;
    const int lr = lane >> 4, lc = lane & 15;
    auto ld_row = [&](const float *__restrict__ p, int x, int nx, int) { // column x of a blurred plane, this lane's row
        const int rc = x < nx ? x : nx - 1;
        const float v = p[(size_t)rc * pt];
        return x < nx ? v : 0.0f;
    };
    float w0[NF], w1[NF], w2[NF], w3[WN], w4[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        w3[j] = j < P ? ld_row(v3, j, w, pt) : 0.0f;
        w4[j] = j < P ? ld_row(v4, j, w, pt) : 0.0f;
        if (FULL) {
            w0[j] = j < P ? ld_row(v0, j, w, pt) : 0.0f;
            w1[j] = j < P ? ld_row(v1, j, w, pt) : 0.0f;
            w2[j] = j < P ? ld_row(v2, j, w, pt) : 0.0f;
        }
    }
    // element e of the ref / dis stream = rows y0 + 4 * (e & 15) + lr, columns 16 * (e >> 4) + lc: a lane loads its
    // {ref, dis} pair with one 8-B load and a row of the block is one whole 128-B line
    auto fetch2 = [&](int e, float &a, float &b) {
        const int x = 16 * (e >> 4) + lc, y = y0 + 4 * (e & 15) + lr;
        const int yc = y < h ? y : h - 1, xc = x < pitch ? x : pitch - 1; // stay inside the plane; such samples are never used
        const float2 v = *(const float2 *)(rdn + 2 * ((size_t)yc * pitch + xc));
        a = v.x; b = v.y;
    };
    auto put = [&](int p, int e, float v) { tile[p][(e >> 4) & 1][4 * (e & 15) + lr][lc] = v; };
    // prologue: block 0 complete in LDS, elements 16 .. 16 + D - 1 in flight
    {
        float a[16], b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) fetch2(i, a[i], b[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) { put(0, i, a[i]); put(1, i, b[i]); }
    }
    float qa[D], qb[D];
#pragma unroll
    for (int i = 0; i < D; ++i) fetch2(16 + i, qa[i], qb[i]);
    __builtin_amdgcn_wave_barrier();
    tmdev::Iir f0 = {0, 0, 0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0, f4 = f0;
    const int T = w + 4;
    for (int t0 = 0; t0 < T; t0 += WN) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int t = t0 + j; // row t lives in slot j, row t-10 in slot (j+P) % WN, which row t+P then takes over
            float s11 = 0.0f, s22 = 0.0f, s12 = 0.0f;
            if (FULL) {
                s11 = tmdev::iir_step(f0, w0[(j + P) % NF] + w0[j % NF]);
                s22 = tmdev::iir_step(f1, w1[(j + P) % NF] + w1[j % NF]);
                s12 = tmdev::iir_step(f2, w2[(j + P) % NF] + w2[j % NF]);
            }
            const float mu1 = tmdev::iir_step(f3, w3[(j + P) % WN] + w3[j]);
            const float mu2 = tmdev::iir_step(f4, w4[(j + P) % WN] + w4[j]);
            if (FULL) {
                w0[(j + P) % NF] = ld_row(v0, t + P, w, pt);
                w1[(j + P) % NF] = ld_row(v1, t + P, w, pt);
                w2[(j + P) % NF] = ld_row(v2, t + P, w, pt);
            }
            w3[(j + P) % WN] = ld_row(v3, t + P, w, pt);
            w4[(j + P) % WN] = ld_row(v4, t + P, w, pt);
            if (t >= 4 && t < T) {
                const int u = t - 4;                 // the column whose maps are evaluated now
                const int slot = (j + WN - 4) % D;   // == u % D because D divides WN
                // element u + 16 (requested D steps ago) into its buffer, element u + 16 + D requested in its place
                __builtin_amdgcn_wave_barrier();
                put(0, u + 16, qa[slot]); put(1, u + 16, qb[slot]);
                fetch2(u + 16 + D, qa[slot], qb[slot]);
                __builtin_amdgcn_wave_barrier();
                const float src = tile[0][(u >> 4) & 1][lane][u & 15], dsv = tile[1][(u >> 4) & 1][lane][u & 15];
                float ssim = 0.0f, art, det;
                if (FULL) tmdev::error_maps(src, dsv, mu1, mu2, s11, s22, s12, ssim, art, det);
                else tmdev::edge_maps(src, dsv, mu1, mu2, art, det);
                if (valid) {
                    float q;
                    if (FULL) { acc[0] += (double)ssim; q = ssim * ssim; q = q * q; acc[3] += (double)q; }
                    acc[1] += (double)art;  q = art * art;   q = q * q; acc[4] += (double)q;
                    acc[2] += (double)det;  q = det * det;   q = q * q; acc[5] += (double)q;
                }
            }
        }
    }
}

// WNF / DF, WNE / DE: register window (rows of the blurred planes in flight = window - 10) and ref/dis queue depth of the
// FULL and the EDGE path.  The engine runs <16, 8, 32, 16> up to 2560 pixels wide and <16, 8, 16, 8> above (measured: FULL 12 ->
// 16 is worth 10 % of this pass at 4K and nothing at 1080p, EDGE 16 -> 32 3.5 % at 1080p and -5 % at 4K); at 217 VGPRs the
// kernel holds 2 waves per SIMD = 8 per CU, one fewer than its LDS would allow, which by itself measured 3.6 % faster.
// PROBE: second instantiation for the placement probe of tm_engine_create (listed apart by profilers, like k_blur_v_jobs').
// grid (slots, jobs.hstart[n]) -- slot-major like the column pass --, block 64.  PART[slot][row block over all jobs][6].
template <int WNF, int DF, int WNE, int DE, int PROBE = 0>
__global__ void __launch_bounds__(64) k_blur_h_jobs_x(TmGeom g, TmJobs jobs, const float *__restrict__ XYB,
                                                      const float *__restrict__ V, double *__restrict__ PART)
{
    __shared__ float tile[2][2][64][17];
    if (jobs.prio > 0) TM_SETPRIO(2);
    const int b = blockIdx.y, slot = blockIdx.x;
    const int j = tm_find_job(jobs.hstart, b);
    const int s = jobs.scale[j], c = jobs.chan[j], mode = jobs.mode[j];
    const TmScaleGeom sg = g.s[s];
    const int y0 = (b - jobs.hstart[j]) * 64;
    const int y = y0 + threadIdx.x;
    const bool valid = y < sg.h;
    const int yy = valid ? y : sg.h - 1;
    const size_t to = sg.off_t + c * sg.plane_t + (size_t)yy;
    const float *rdn = XYB + (size_t)slot * 2 * g.pyr + 2 * (sg.off + c * sg.plane);
    const float *v0 = V + (size_t)(slot * 5 + 0) * g.pyr_t + to;
    const float *v1 = V + (size_t)(slot * 5 + 1) * g.pyr_t + to;
    const float *v2 = V + (size_t)(slot * 5 + 2) * g.pyr_t + to;
    const float *v3 = V + (size_t)(slot * 5 + 3) * g.pyr_t + to;
    const float *v4 = V + (size_t)(slot * 5 + 4) * g.pyr_t + to;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (mode == TM_MODE_FULL) blur_h_job_x<true, WNF, DF>(tile, rdn, v0, v1, v2, v3, v4, y0, sg.w, sg.h, sg.pitch, sg.pitch_t, valid, acc);
    else blur_h_job_x<false, WNE, DE>(tile, rdn, v0, v1, v2, v3, v4, y0, sg.w, sg.h, sg.pitch, sg.pitch_t, valid, acc);
    if (tm_wave_sum6(acc)) {
        double *o = PART + ((size_t)slot * jobs.hstart[TM_MAX_JOBS] + b) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = acc[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Row pass for SMALL launches ("split"): the same arithmetic and the same per-lane order of accumulation as k_blur_h_jobs_x -- its
// PART entries are bit-identical --, cut across EIGHT waves per 64-row block.  Why: a launch of a few pairs leaves most of the chip
// idle, and what its row pass takes is the time ONE row block needs to walk its w + 4 steps.  Round 3 gave a block three, then five
// waves (0.58 -> 0.35 -> 0.26 ms per 1080p row); round 4 measured what bounds it (profiles/r04c_*_sq_summary.txt, r04d ... r04j):
//   * a step of the FULL path is ~180 instructions over all its waves (five recurrences of ~20 with their loads and ring stores, ~10 for
//     the ref / dis blocks, 37 for the ssim map and its sums, 32 for the edge maps and theirs), a SIMD issues one instruction per four
//     cycles, and a workgroup lives on ONE CU with its four SIMDs: 180 instructions / 4 SIMDs x 4 cycles = 180 cycles per step if the
//     waves were spread evenly, 215-225 measured (1 924 steps of a 1080p row: 0.175 ms) -- the time per step is what the CU can issue
//     for the block, not any single wave's count: a consumer cut into two waves of ~20 (ten waves per block, tried) changed nothing,
//     nor did producers without clamps (13 instead of 20);
//   * the consumers must not wait per step: an `if (valid)` around every step put each into its own exec-masked block, so that no LDS
//     read of step j + 1 could be issued before the arithmetic of step j (~340 cycles per step: 0.26 -> 0.20 ms without it);
//   * the ref / dis blocks need more than one phase of look-ahead: with one, a phase of 16 steps cannot be shorter than a load from
//     HBM on an idle chip (8 pairs per launch: row pass 0.34 -> 0.28 ms with two).
// Roles:  waves 0, 1, 2  sigma11, sigma22, sigma12 (FULL jobs)     waves 3, 4  mu1, mu2
//         wave 5  the ref / dis blocks: fetched in the normal orientation two phases ahead, parked transposed in LDS (this replaces the
//                 reference's nppiTranspose)
//         wave 6  the ssim map and its two sums (FULL)              wave 7  the two edge maps and their four sums
// (an order that pairs the heavy consumers with the light waves under "wave i runs on SIMD i % 4" -- 47 / 52 / 40 / 40 instead of 40 / 30 /
// 57 / 52 instructions per step and SIMD -- measured SLOWER: one pair 0.175 -> 0.196 ms, eight pairs 0.285 -> 0.298; the placement of a
// workgroup's waves is not that simple, and this order is the measured best)
// Steps run in phases of 16 with one LDS barrier per phase: the producers fill half (phase & 1) of a two-phase ring [2][16][5][64]
// (40 KB) while the consumers empty the other half -- step t of a recurrence emits column t - 4, stored at ring position t --; a
// producer keeps a 32-row register window of its plane (22 rows of loads in flight); ref / dis block b (columns 16 b .. + 15) is
// requested during phase b - 2, written during phase b, read during phases b + 1 and b + 2: four tile buffers (35 KB).  75 KB of LDS
// and 112 registers: two workgroups per CU.  The engine runs it up to ~2 600 row blocks per launch (32 pairs of 1080p beside the fused
// kernel; above, the one-wave pass has enough waves to hide its latencies); TM_VARIANT_SPLIT_ROWS / _WHOLE_ROWS force / forbid it.
// Measured (1080p, pairs per launch, pairs/s; round 3 -> round 4): 1: 2.4 k -> 3.0 k, 4: 6.6 k -> 8.4 k, 8: 10.0 k -> 11.9 k,
// 16: 11.7 k -> 12.8 k, 32: 12.5 k -> 13.5 k.  grid (slots, jobs.hstart[n]), block 512.
// ------------------------------------------------------------------------------------------------
#define TM_SPLIT_WAVES 8
__device__ __forceinline__ void blur_h_split_producer(float (*__restrict__ ring)[16][5][64], const float *__restrict__ v, int plane, int w, int pt, int nphases)
{
    // v: this lane's row of a transposed blurred plane (column x at v[x * pt]); step t emits column t - 4 into ring[phase & 1][t & 15][plane]
    constexpr int WN = 32, P = WN - 10;
    const int lane = threadIdx.x & 63;
    auto ld_col = [&](int x) { // column x, this lane's row; 0 outside
        const int rc = x < w ? x : w - 1;
        const float val = v[(size_t)rc * pt];
        return x < w ? val : 0.0f;
    };
    float win[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) win[j] = j < P ? ld_col(j) : 0.0f;
    tmdev::Iir f = {0, 0, 0, 0, 0, 0};
    for (int ph0 = 0; ph0 < nphases; ph0 += 2) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ph = ph0 + sub;
            if (ph >= nphases) break; // (the same for every wave of the workgroup: they all meet at the same barriers)
            // (a second copy of this loop without clamp, 64-bit multiply and select for interior phases -- 14 instead of 20 instructions per
            // step -- was measured twice: the producers' work went from 97 to 90 cycles per step and the pass did not move, 0.175 -> 0.18 ms: it is
            // the consumers that every phase waits for, profiles/r04s_split_timing.log)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = 16 * ph + j, sl = 16 * sub + j; // row t lives in slot t % 32 (ph0 is even), row t - 10 in slot (sl + P) % 32, which row t + P then takes over
                const float o = tmdev::iir_step(f, win[(sl + P) % WN] + win[sl]);
                win[(sl + P) % WN] = ld_col(t + P);
                ring[ph & 1][j][plane][lane] = o;
            }
            TM_LDS_BARRIER();
        }
    }
}

// the ref / dis blocks: block e = rows y0 + 4 i + (lane >> 4) (i = 0 .. 15), columns 16 e + (lane & 15) -- one 8-byte load per lane and
// row group, a row of the block is one whole 128-B line --, requested D phases before it is written into its tile buffer
template <int D>
__device__ __forceinline__ void blur_h_split_fetcher(float (*__restrict__ tile)[4][64][17], const float *__restrict__ rdn, int y0, int h, int pitch, int nphases)
{
    const int lane = threadIdx.x & 63;
    const int lr = lane >> 4, lc = lane & 15;
    float qa[D][16], qb[D][16];
    auto fetch_block = [&](int e, float (&a)[16], float (&b)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int x = 16 * e + lc, y = y0 + 4 * i + lr;
            const int yc = y < h ? y : h - 1, xc = x < pitch ? x : pitch - 1; // stay inside the plane; such samples are never used
            const float2 val = *(const float2 *)(rdn + 2 * ((size_t)yc * pitch + xc));
            a[i] = val.x; b[i] = val.y;
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch_block(d, qa[d], qb[d]);
    for (int ph0 = 0; ph0 < nphases; ph0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ph = ph0 + d;
            if (ph >= nphases) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) { tile[0][ph & 3][4 * i + lr][lc] = qa[d][i]; tile[1][ph & 3][4 * i + lr][lc] = qb[d][i]; }
            fetch_block(ph + D, qa[d], qb[d]);
            TM_LDS_BARRIER();
        }
    }
}

// SSIM: the ssim map and its two sums (all five blurred values); otherwise the two edge maps and their four sums (mu1, mu2, ref, dis).
// The per-lane order of accumulation is that of k_blur_h_jobs_x.  Runs one phase behind the producers.
template <bool SSIM>
__device__ __forceinline__ void blur_h_split_consumer(const float (*__restrict__ ring)[16][5][64], const float (*__restrict__ tile)[4][64][17],
                                                      int w, bool valid, int nphases, double (&acc)[6])
{
    const int lane = threadIdx.x & 63;
    const int T = w + 4;
    // Lanes of rows below the image accumulate too (their inputs are the clamped last row's: finite) and are zeroed at the end; the LDS
    // reads of a block of eight steps are issued together, in front of its arithmetic.
    struct In { float mu1, mu2, s11, s22, s12, src, dsv; };
    auto fetch = [&](int ph1, int j, int t) __attribute__((always_inline)) {
        const int u = t - 4; // the column whose maps are evaluated at step t
        const float (*r)[64] = ring[ph1 & 1][j];
        In v;
        v.mu1 = r[3][lane]; v.mu2 = r[4][lane];
        v.s11 = v.s22 = v.s12 = 0.0f; v.src = v.dsv = 0.0f;
        if (SSIM) { v.s11 = r[0][lane]; v.s22 = r[1][lane]; v.s12 = r[2][lane]; }
        else { v.src = tile[0][(u >> 4) & 3][lane][u & 15]; v.dsv = tile[1][(u >> 4) & 3][lane][u & 15]; }
        return v;
    };
    auto step = [&](const In &v) __attribute__((always_inline)) {
        float ssim = 0.0f, art = 0.0f, det = 0.0f, q;
        if (SSIM) { // (error_maps evaluates all three maps; the compiler drops the half whose results are not used)
            tmdev::error_maps(v.src, v.dsv, v.mu1, v.mu2, v.s11, v.s22, v.s12, ssim, art, det);
            acc[0] += (double)ssim; q = ssim * ssim; q = q * q; acc[3] += (double)q;
        } else {
            tmdev::edge_maps(v.src, v.dsv, v.mu1, v.mu2, art, det);
            acc[1] += (double)art;  q = art * art;   q = q * q; acc[4] += (double)q;
            acc[2] += (double)det;  q = det * det;   q = q * q; acc[5] += (double)q;
        }
    };
    for (int ph = 0; ph < nphases; ++ph) {
        if (ph > 0) {
            const int tb = 16 * (ph - 1);
            if (tb >= 4 && tb + 15 < T) { // every step of the phase emits a column (all phases but the first and the last one or two): straight-line code
#pragma unroll
                for (int j0 = 0; j0 < 16; j0 += 8) {
                    In v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fetch(ph - 1, j0 + j, tb + j0 + j);
#pragma unroll
                    for (int j = 0; j < 8; ++j) step(v[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (tb + j >= 4 && tb + j < T) step(fetch(ph - 1, j, tb + j));
            }
        }
        TM_LDS_BARRIER();
    }
    if (!valid) {
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = 0.0;
    }
}

// a wave with nothing to do in this job: it only keeps the workgroup's barriers company
__device__ __forceinline__ void blur_h_split_idle(int nphases)
{
    for (int ph = 0; ph < nphases; ++ph) TM_LDS_BARRIER();
}

__global__ void __launch_bounds__(64 * TM_SPLIT_WAVES) k_blur_h_jobs_split(TmGeom g, TmJobs jobs, const float *__restrict__ XYB, const float *__restrict__ V,
                                                                           double *__restrict__ PART)
{
    __shared__ float ring[2][16][5][64];
    __shared__ float tile[2][4][64][17];
    const int b = blockIdx.y, slot = blockIdx.x;
    const int j = tm_find_job(jobs.hstart, b);
    const int s = jobs.scale[j], c = jobs.chan[j], mode = jobs.mode[j];
    const TmScaleGeom sg = g.s[s];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform
    const int lane = threadIdx.x & 63;
    const int y0 = (b - jobs.hstart[j]) * 64;
    const int y = y0 + lane;
    const bool valid = y < sg.h;
    const int yy = valid ? y : sg.h - 1;
    const size_t to = sg.off_t + c * sg.plane_t + (size_t)yy;
    const float *rdn = XYB + (size_t)slot * 2 * g.pyr + 2 * (sg.off + c * sg.plane);
    const int nphases = (sg.w + 4 + 15) / 16 + 1; // the consumers run one phase behind the producers
    const bool full = mode == TM_MODE_FULL;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    int mine = 0; // bits: which maps' sums this wave holds at the end (1 ssim, 2 edge)
    if (wave == 6) { // the ssim map and its sums (FULL)
        if (full) { blur_h_split_consumer<true>(ring, tile, sg.w, valid, nphases, acc); mine = 1; }
        else blur_h_split_idle(nphases);
    } else if (wave == 7) { // the edge maps and their sums
        blur_h_split_consumer<false>(ring, tile, sg.w, valid, nphases, acc); mine = 2;
    } else if (wave == 5) { // the ref / dis blocks, two phases ahead (three would cost the second workgroup per CU: 138 registers)
        blur_h_split_fetcher<2>(tile, rdn, y0, sg.h, sg.pitch, nphases);
    } else if (wave >= 3 || full) { // producer of plane `wave`
        blur_h_split_producer(ring, V + (size_t)(slot * 5 + wave) * g.pyr_t + to, wave, sg.w, sg.pitch_t, nphases);
    } else blur_h_split_idle(nphases); // the sigma producers of an EDGE job
    if (mine == 0) return;
    // a consumer holds some of the six sums of the row block (zeros elsewhere); the shuffle tree adds every entry in the order
    // k_blur_h_jobs_x adds it, so the entries come out bit-identical.  An EDGE job has no ssim sums: zeros, as k_blur_h_jobs_x writes
    const bool w_ssim = (mine & 1) || !full, w_edge = (mine & 2) != 0;
    if (tm_wave_sum6(acc)) {
        double *o = PART + ((size_t)slot * jobs.hstart[TM_MAX_JOBS] + b) * 6;
        if (w_ssim) { o[0] = acc[0]; o[3] = acc[3]; }
        if (w_edge) { o[1] = acc[1]; o[2] = acc[2]; o[4] = acc[4]; o[5] = acc[5]; }
    }
}

// ------------------------------------------------------------------------------------------------
// EDGE jobs in ONE kernel ("fused"): column recurrence, row recurrence, the edge half of compute_error_maps and its four sums
// without the pass-1 arena.  An EDGE job (only mu1, mu2 carry weight: scale 0 of X and B = half of all pixels) costs the two
// passes above 8 units of HBM traffic per pixel-channel (2 R + 2 W, then 4 R); here it costs the 2 units of its input.
//   * one wave = one BAND of 32 image rows of one (slot, job), walking right in tiles of 32 columns; a lane is a
//     (column, side) pair during the column phase of a tile and a (row, side) pair during its row phase -- ref and dis run the
//     same recurrence, so both phases are plain f32 code on 64 full lanes;
//   * column phase of a tile: the 42 input rows it needs (32 + the 10 rows of history; window slot k = image row 32 band - 6 + k)
//     were requested a whole row phase earlier (one 256-B run per row and wave); 32 steps of the recurrence whose STATE comes from
//     the band above (below: hand-off); every step parks {V, original} of its row in the wave's LDS tile;
//   * row phase: lane (row, side) walks the 32 columns of the tile, one 8-byte read per step; it emits column x - 4 at step x;
//     recurrence state, the last ten V and the last four originals of its row stay in registers from tile to tile;
//     e = 1 + |orig - mu| on each side, both e of the pixel through two DPP broadcasts, d1 = fma(e_dis, 1 / e_ref, -1) on both
//     lanes of the pair, of which the ref lane accumulates the artifact sums and the dis lane the detail_loss sums, f64, in
//     column order: exactly the per-row sums of k_blur_h_jobs_x.  They go to EROWS and
//     k_finish_edge adds the rows of a 64-row block in the order of tm_wave_sum6 -> the PART entries, and everything after them,
//     are bit-identical with the two-pass kernels (the GPU tier checks that);
//   * hand-off: the six state values of a (column, side) after the last step of band b are what band b + 1 starts from.  They
//     travel through HS as 8-byte {value, tag} words written and read with device-scope atomic accesses (tag = launch epoch and
//     band: a reader spins until all six words carry the tag it expects; no fence, no flag: the word is its own flag), two
//     buffers per plane alternate by band parity (band b + 2 cannot overwrite what band b + 1 still has to read: it needs band
//     b + 1's state of that tile first).  Work is handed out plane-major by TICKET (an atomic counter drawn at workgroup
//     start; ticket = band * workgroups per band + planes): a band only ever waits for the band above, whose ticket is
//     smaller and therefore held by a workgroup that has started -- no deadlock whatever the dispatch order --, and since
//     all planes start band b before any starts b + 1 the producer is normally tiles ahead (the wait costs 3 % of the kernel).  A wait that lasts longer than ~2^22 polls sets *status (the host reports TM_ERR_HIP).
// LDS: 32 rows x 132 floats (32 columns x 2 sides x 2, + 4: eight rows cover the 32 banks in 8-byte reads) = 16.9 KB per
// wave.  NW = 4 waves form a workgroup, each with its own band and tile: the hardware spreads the waves of one workgroup over
// the four SIMDs of its CU (two such workgroups per CU = two waves per SIMD), while single-wave workgroups land 2 / 3 / 4 to a
// SIMD and the waves of the crowded SIMDs take 1.4-1.9 x as long (measured: 0.98 vs 1.09 ms per 64 1080p pairs).  GROUPED (what
// the engine launches): the four waves are four ADJACENT bands of one plane, and the state crosses the three boundaries inside
// the group through an LDS mailbox (TmEfMailbox) -- PMC: 4.4 GB of HBM traffic per 64 1080p pairs with every boundary through
// memory (2.1 GB of input, 0.7 GB of it read twice -- 10 of the 42 window rows --, 0.8 GB of words written and as many read).  What bounds
// the kernel is what a SIMD can issue: ~1 550 instructions per tile (row phase 970: 12 for the recurrence, 8 for the division,
// 4 binary64 ones for the two sums, per step), ~3 300 VALU-pipe cycles; one wave alone on a SIMD walks its band of a 1080p plane
// in 0.28 ms, two share the SIMD at 0.31 / 0.45 ms (the older wave wins the arbitration).
// grid: one workgroup per ticket (ceil(planes / NW) * bands) when the kernel has the chip to itself; beside the two blur passes a
// persistent launch of 7/8 of a workgroup per CU that share the tickets (tm_engine.hip).  block 64 NW; plane = slot * ne + job.
// HS[plane][2][hs_tiles][6][64], EROWS[plane][er_bands][64][2].
// ------------------------------------------------------------------------------------------------
#define TM_EF_S 132
// (tm_ll_store / tm_ll_load -- the tagged 64-bit hand-off words, relaxed agent-scope atomics -- and TM_WAVE_ALL: tm_platform.h)

// what the fused kernel needs to know about its jobs (a small kernarg: the tile loop keeps its scalars in registers)
struct TmEdgeJob {
    int w, h;
    int rowf;                   // floats per row of the interleaved plane (2 * pitch)
    int part0;                  // first PART row block of the job (jobs.hstart[j])
    unsigned long long in_off;  // float offset of the job's plane inside a slot's interleaved pyramid
};
struct TmEdgeArgs {
    TmEdgeJob job[TM_MAX_JOBS];
    unsigned long long slot_stride; // floats per slot of the interleaved pyramid (2 * g.pyr)
    int ne, hs_tiles, er_bands, part_stride /* PART row blocks per slot */;
};
static inline void tm_make_edge_args(TmEdgeArgs *a, const TmGeom *g, const TmJobs *jobs, int hs_tiles, int er_bands)
{
    a->ne = jobs->n - jobs->nfull; a->hs_tiles = hs_tiles; a->er_bands = er_bands; a->part_stride = jobs->hstart[TM_MAX_JOBS];
    a->slot_stride = 2 * g->pyr;
    for (int e = 0; e < TM_MAX_JOBS; ++e) {
        const int j = jobs->nfull + (e < a->ne ? e : 0);
        const TmScaleGeom *sg = &g->s[jobs->scale[j]];
        a->job[e].w = sg->w; a->job[e].h = sg->h; a->job[e].rowf = 2 * sg->pitch; a->job[e].part0 = jobs->hstart[j];
        a->job[e].in_off = 2 * (sg->off + (unsigned long long)jobs->chan[j] * sg->plane);
    }
}

// one pixel of the edge maps on a (row, side) lane pair (error_maps.rs:45-59, the same operations as tmdev::edge_maps):
//   * e = 1 + |orig - mu| on each lane; both lanes of the pair then hold e_ref (the even lane's) and e_dis (the odd lane's): two
//     DPP quad broadcasts;
//   * 1 / e_ref as the compiler's own IEEE division sequence without v_div_scale / v_div_fixup (1 <= e_ref, far from the ends of
//     the exponent range: the same operations on the same values, like ssim_div; the numerator 1 makes the first product exact);
//   * the ref lane keeps artifact = max(d1, 0), the dis lane detail_loss = max(-d1, 0): the sign flips by a per-lane mask and the
//     maximum with 0 is an integer maximum (a float below or equal to zero is a negative integer or 0; no NaN can occur);
//   * lanes of rows below the image accumulate whatever they compute: their sums are dropped at the end.
__device__ __forceinline__ void ef_accumulate(float og, float mu, bool dis, unsigned sgn, double &a1, double &a4)
{
    const float e = 1.0f + fabsf(og - mu);
    float e_ref, e_dis;
    tm_pair_values(e, dis, e_ref, e_dis);           // the even lane of a pair is the ref side, the odd lane the dis side
    const float denom = tm_rcp_inrange(e_ref);      // 1 / (1 + |source - mu1|)
    const float d1 = __builtin_fmaf(e_dis, denom, -1.0f);                                                                    // numer = 1 + |distorted - mu2|
    const int bits = (int)(__float_as_uint(d1) ^ sgn);
    const float m = __uint_as_float((unsigned)(bits > 0 ? bits : 0));
    a1 += (double)m;
    float q = m * m; q = q * q;
    a4 += (double)q;
}

template <bool GUARD>
__device__ __forceinline__ void ef_row_phase(const float *__restrict__ trow, int n0, int w, tmdev::Iir &fr, float (&vc)[10], float (&oc)[4],
                                             bool dis, unsigned sgn, double &a1, double &a4)
{
    // trow: this lane's tile row + 2 * side; n0 = the column emitted at step 0 (32 tile - 4)
    float v[32], o[32];
#pragma unroll
    for (int x = 0; x < 32; ++x) {
        const tm_g2 pr = *(const tm_g2 *)(trow + 4 * x); // {V, original} of column 32 tile + x
        v[x] = pr.x; o[x] = pr.y;
        const float mu = tmdev::iir_step(fr, (x >= 10 ? v[x >= 10 ? x - 10 : 0] : vc[x < 10 ? x : 0]) + pr.x);
        if (!GUARD || (n0 + x >= 0 && n0 + x < w)) ef_accumulate(x >= 4 ? o[x >= 4 ? x - 4 : 0] : oc[x & 3], mu, dis, sgn, a1, a4);
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) vc[k] = v[22 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) oc[k] = o[28 + k];
}

// GROUPED launches: the four waves of a workgroup take four vertically adjacent bands of ONE plane, and the column recurrence's
// state crosses the three band boundaries inside the group through LDS -- a two-slot mailbox per boundary, `full` / `done`
// counters (tile number + 1) as flow control -- instead of through memory: a quarter of the hand-off words (and their polls) reach
// HBM, and the ten input rows two neighbouring bands share are read by two waves of one CU within microseconds of each other.
#define TM_EF_MS 2
struct TmEfMailbox {
    float v[3][TM_EF_MS][6][64];
    unsigned full[3][TM_EF_MS], done[3][TM_EF_MS];
};
// (TM_EF_SPIN_PAUSE: s_sleep 1; TM_EF_LDS_FENCE: a workgroup-scope fence -- tm_platform.h)

// one band of one plane (see above); tile: this wave's LDS tile
template <bool GROUPED>
__device__ __forceinline__ void ef_band(float *__restrict__ tile, TmEfMailbox *__restrict__ mb, int wv, const TmEdgeArgs &A, int p, int band, int planes,
                                        const float *__restrict__ XYB, unsigned long long *__restrict__ HS, unsigned epoch, double *__restrict__ EROWS,
                                        int *__restrict__ status, int dbg)
{
    const int slot = p / A.ne;
    const TmEdgeJob J = A.job[p - slot * A.ne];
    const int h = J.h, w = J.w;
    const int nbands = (h + 31) >> 5, ntiles = (w + 31) >> 5;
    if (band >= nbands) return;
    const int lane = threadIdx.x & 63, cl = lane >> 1, side = lane & 1;
    const unsigned tag_in = (epoch << 8) | (unsigned)((band - 1) & 255), tag_out = (epoch << 8) | (unsigned)(band & 255);
    const int y0 = 32 * band - 6;                        // image row of window slot 0
    const bool interior = y0 >= 0 && y0 + 41 < h;        // every window row exists
    const unsigned rowb = (unsigned)J.rowf * 4u;         // bytes per interleaved row; a plane stays below 2^31 bytes (16 384 x 16 384 x 8)
    // window loads take ONE scalar base and a 32-bit lane offset each (global_load ... s[base], one v_add per row): interior
    // bands from row y0 on, the first / last bands from the top of the plane with clamped row numbers (and zeros afterwards)
    TM_GLOBAL_AS const char *inb = (TM_GLOBAL_AS const char *)tm_uniform_ptr(XYB + (size_t)slot * A.slot_stride + J.in_off + (interior ? (size_t)y0 * J.rowf : 0));
    const unsigned long long *hs_in = HS + ((size_t)p * 2 + ((band - 1) & 1)) * A.hs_tiles * 384 + lane;
    unsigned long long *hs_out = HS + ((size_t)p * 2 + (band & 1)) * A.hs_tiles * 384 + lane;
    // where the state comes from and goes to: memory (tagged words) across groups, the LDS mailbox inside a group
    const bool in_mem = band > 0 && (!GROUPED || wv == 0), in_lds = GROUPED && wv > 0;
    const bool out_mem = band + 1 < nbands && (!GROUPED || wv == 3), out_lds = GROUPED && wv < 3 && band + 1 < nbands;
    const bool valid = 32 * band + cl < h, dis = side != 0;
    const unsigned sgn = (unsigned)side << 31;
    float *tcol = tile + 4 * cl + 2 * side;                 // column phase: + row * TM_EF_S
    const float *trow = tile + cl * TM_EF_S + 2 * side;     // row phase: + 4 * step

    float win[42];
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    auto fetch = [&](int i) {
        const unsigned xb = (unsigned)min(32 * i + cl, w - 1) * 8u + (unsigned)side * 4u; // lanes past the right edge shadow the last column (and are zeroed below)
        if (interior) {
#pragma unroll
            for (int k = 0; k < 42; ++k) win[k] = *(TM_GLOBAL_AS const float *)(inb + (xb + (unsigned)k * rowb));
        } else {
#pragma unroll
            for (int k = 0; k < 42; ++k) {
                const int y = y0 + k, yc = y < 0 ? 0 : (y < h ? y : h - 1);
                win[k] = *(TM_GLOBAL_AS const float *)(inb + (xb + (unsigned)yc * rowb));
            }
        }
        if (in_mem) {
#pragma unroll
            for (int k = 0; k < 6; ++k) st[k] = tm_ll_load(hs_in + (size_t)i * 384 + k * 64);
        }
    };
    fetch(0);
    tmdev::Iir fr = {0, 0, 0, 0, 0, 0};
    float vc[10], oc[4];
#pragma unroll
    for (int k = 0; k < 10; ++k) vc[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) oc[k] = 0.0f;
    double a1 = 0.0, a4 = 0.0;
    for (int i = 0; i < ntiles; ++i) {
        const bool colok = 32 * i + cl < w;
        tmdev::Iir f = {0, 0, 0, 0, 0, 0};
        if (in_mem) {
            int polls = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 6; ++k) ok = ok && (unsigned)(st[k] >> 32) == tag_in;
                if (TM_WAVE_ALL(ok) || (dbg & 1)) break;
                if (++polls > ((dbg & 2) ? (1 << 10) : (1 << 22)) || *(volatile int *)status) { *(volatile int *)status = 1; break; } // (fault injection gives up soon: the test does not wait seconds)
                TM_SLEEP(8);
#pragma unroll
                for (int k = 0; k < 6; ++k) st[k] = tm_ll_load(hs_in + (size_t)i * 384 + k * 64);
            }
            f.p1a = __uint_as_float((unsigned)st[0]); f.p1b = __uint_as_float((unsigned)st[1]); f.p1c = __uint_as_float((unsigned)st[2]);
            f.p2a = __uint_as_float((unsigned)st[3]); f.p2b = __uint_as_float((unsigned)st[4]); f.p2c = __uint_as_float((unsigned)st[5]);
        }
        if (GROUPED && in_lds) { // the band above is wave wv - 1 of this workgroup
            const int sl = i % TM_EF_MS;
            int polls = 0;
            while (*(volatile unsigned *)&mb->full[wv - 1][sl] != (unsigned)i + 1u) {
                if (++polls > (1 << 24) || ((polls & 1023) == 0 && *(volatile int *)status)) { *(volatile int *)status = 1; break; } // (the status word lives in memory: looked at now and then)
                TM_EF_SPIN_PAUSE();
            }
            TM_EF_LDS_FENCE();
            const float (*src)[64] = mb->v[wv - 1][sl];
            f.p1a = src[0][lane]; f.p1b = src[1][lane]; f.p1c = src[2][lane]; f.p2a = src[3][lane]; f.p2b = src[4][lane]; f.p2c = src[5][lane];
            TM_EF_LDS_FENCE();
            __builtin_amdgcn_wave_barrier(); // (the 64 lanes are one instruction stream: every lane has read before lane 0 says so)
            if (lane == 0) *(volatile unsigned *)&mb->done[wv - 1][sl] = (unsigned)i + 1u;
        }
        if (!interior) { // rows above / below the image are zeros (blur.rs:104-110); here, not in fetch: a select behind every load would wait for it
#pragma unroll
            for (int k = 0; k < 42; ++k) win[k] = (y0 + k >= 0 && y0 + k < h) ? win[k] : 0.0f;
        }
        if (32 * i + 31 >= w) { // the last tile: columns past the right edge enter both recurrences as zeros (blur.rs:104-110)
#pragma unroll
            for (int k = 0; k < 42; ++k) win[k] = colok ? win[k] : 0.0f;
        }
        if (band == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) (void)tmdev::iir_step(f, 0.0f + win[6 + t]); // steps 0..3 of the recurrence: no output row yet
        }
        // ---- column phase: step j = image step 32 band + 4 + j reads rows win[j + 10] and win[j] and emits row 32 band + j
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const float o = tmdev::iir_step(f, win[jj] + win[jj + 10]);
            tcol[jj * TM_EF_S] = o;
            tcol[jj * TM_EF_S + 1] = win[jj + 6];
        }
        if (out_mem && !(dbg & 2)) {
            unsigned long long *o = hs_out + (size_t)i * 384;
            tm_ll_store(o, f.p1a, tag_out); tm_ll_store(o + 64, f.p1b, tag_out); tm_ll_store(o + 128, f.p1c, tag_out);
            tm_ll_store(o + 192, f.p2a, tag_out); tm_ll_store(o + 256, f.p2b, tag_out); tm_ll_store(o + 320, f.p2c, tag_out);
        }
        if (GROUPED && out_lds) { // the band below is wave wv + 1 of this workgroup: wait until it has taken what this slot held
            const int sl = i % TM_EF_MS;
            int polls = 0;
            while (i >= TM_EF_MS && *(volatile unsigned *)&mb->done[wv][sl] != (unsigned)(i - TM_EF_MS) + 1u) {
                if (++polls > (1 << 24) || ((polls & 1023) == 0 && *(volatile int *)status)) { *(volatile int *)status = 1; break; } // (the status word lives in memory: looked at now and then)
                TM_EF_SPIN_PAUSE();
            }
            TM_EF_LDS_FENCE();
            float (*dst)[64] = mb->v[wv][sl];
            dst[0][lane] = f.p1a; dst[1][lane] = f.p1b; dst[2][lane] = f.p1c; dst[3][lane] = f.p2a; dst[4][lane] = f.p2b; dst[5][lane] = f.p2c;
            TM_EF_LDS_FENCE();
            __builtin_amdgcn_wave_barrier(); // (every lane has written before lane 0 says so)
            if (lane == 0) *(volatile unsigned *)&mb->full[wv][sl] = (unsigned)i + 1u;
        }
        __builtin_amdgcn_wave_barrier();
        fetch(min(i + 1, ntiles - 1)); // requested now, lands during the row phase (past the last tile: that tile again, never used)
        // ---- row phase: step x = image step 32 i + x of this lane's row, emits column 32 i + x - 4
        if (i == 0 || 32 * i + 27 >= w) ef_row_phase<true>(trow, 32 * i - 4, w, fr, vc, oc, dis, sgn, a1, a4);
        else ef_row_phase<false>(trow, 32 * i - 4, w, fr, vc, oc, dis, sgn, a1, a4);
        __builtin_amdgcn_wave_barrier();
    }
    // the last four steps of the row recurrence (image steps 32 ntiles .. + 3, input 0): columns up to w - 1
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float mu = tmdev::iir_step(fr, vc[x] + 0.0f);
        if (32 * ntiles - 4 + x < w) ef_accumulate(oc[x], mu, dis, sgn, a1, a4);
    }
    double *er = EROWS + (((size_t)p * A.er_bands + band) * 64 + lane) * 2;
    er[0] = valid ? a1 : 0.0; er[1] = valid ? a4 : 0.0; // rows below the image: nothing (the two-pass kernels never add them)
}

template <int NW, bool GROUPED = false>
__global__ void __launch_bounds__(64 * NW) TM_WAVES_PER_SIMD(2) k_blur_edge_fused(TmEdgeArgs A, int planes, int groups, unsigned total, const float *__restrict__ XYB,
                                                                               unsigned long long *__restrict__ HS, const unsigned *__restrict__ epoch_p,
                                                                               unsigned *__restrict__ ticket, double *__restrict__ EROWS, int *__restrict__ status, int dbg = 0)
{
    // NW waves per workgroup, each with its own band and tile: the hardware spreads the waves of ONE workgroup over the SIMDs of
    // its CU, single-wave workgroups land 2 / 3 / 4 to a SIMD.  !GROUPED: the same band of NW planes (ticket = band * groups +
    // plane group); GROUPED (NW = 4): four adjacent bands of one plane (ticket = band group * planes + plane).
    static_assert(!GROUPED || NW == 4, "grouped launches: four bands per workgroup");
    __shared__ __attribute__((aligned(16))) float tiles[NW][32 * TM_EF_S];
    __shared__ typename std::conditional<GROUPED, TmEfMailbox, unsigned>::type mbox_s; // (not GROUPED: a placeholder that is never read)
    TmEfMailbox *const mbox = (TmEfMailbox *)&mbox_s;
    __shared__ unsigned s_ticket;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned epoch = *epoch_p;
    // which (planes, band) a workgroup runs is decided by a TICKET it draws, not by its block index: whoever holds ticket t knows
    // that every smaller ticket is held by a workgroup that is already running (or done), whatever order the hardware dispatches
    // the grid in -- the bands above, which this one waits for, have smaller tickets.  A workgroup draws tickets until none is
    // left: `total` workgroups run one each, fewer (a persistent launch) share them.
    for (;;) {
        __syncthreads(); // everybody has read the previous ticket (and is done with the mailbox)
        if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
        if (GROUPED && threadIdx.x < 3 * TM_EF_MS) { (&mbox->full[0][0])[threadIdx.x] = 0u; (&mbox->done[0][0])[threadIdx.x] = 0u; }
        __syncthreads();
        const unsigned tk = s_ticket;
        if (tk >= total) break;
        const int p = GROUPED ? (int)(tk % (unsigned)planes) : (int)(tk % (unsigned)groups) * NW + wv;
        const int band = GROUPED ? (int)(tk / (unsigned)planes) * 4 + wv : (int)(tk / (unsigned)groups);
        if (p < planes) ef_band<GROUPED>(tiles[wv], mbox, wv, A, p, band, planes, XYB, HS, epoch, EROWS, status, dbg);
    }
}

// the rows of an EDGE job -> the PART entries k_blur_h_jobs_x would have written: per 64-row block, sum over rows in the order of
// tm_wave_sum6 (row i with row i + 32, then the shuffle tree 16, 8, 4, 2, 1), kinds 1, 4 (artifact: ref lanes) and 2, 5 (detail_loss:
// dis lanes).  Also advances the launch epoch of the hand-off tags.  grid (slots * ne), block 64: thread = (row block, kind).
__global__ void __launch_bounds__(64) k_finish_edge(TmEdgeArgs A, const double *__restrict__ EROWS, double *__restrict__ PART, unsigned *__restrict__ epoch_p)
{
    const int p = blockIdx.x, slot = p / A.ne, er_bands = A.er_bands;
    const TmEdgeJob J = A.job[p - slot * A.ne];
    const int h = J.h, nbands = (h + 31) >> 5, nblk = (h + 63) >> 6;
    for (int item = threadIdx.x; item < nblk * 4; item += 64) {
        const int blk = item >> 2, q = item & 3, side = q & 1, pw = q >> 1; // q: 0 art, 1 det, 2 art^4, 3 det^4
        double a[64];
        for (int i = 0; i < 64; ++i) {
            const int band = 2 * blk + (i >> 5);
            a[i] = band < nbands ? EROWS[(((size_t)p * er_bands + band) * 64 + 2 * (i & 31) + side) * 2 + pw] : 0.0;
        }
        // the order in which tm_wave_sum6 adds the 64 lanes of a row block in k_blur_h_jobs_x: the entries come out bit-identical
        for (int off = 32; off > 0; off >>= 1)
            for (int i = 0; i < off; ++i) a[i] += a[i + off];
        const double tot = a[0];
        PART[((size_t)slot * A.part_stride + J.part0 + blk) * 6 + 1 + side + 3 * pw] = tot;
    }
    if (p == 0 && threadIdx.x == 0) { // next launch: new tags (the tag holds 24 bits of the epoch; 0 is never used: memory starts out as zeros), tickets from 0
        const unsigned next = (*epoch_p + 1u) & 0xFFFFFFu;
        *epoch_p = next ? next : 1u;
        epoch_p[1] = 0u;
    }
}

// the accumulator reset of a launch as a kernel: a memset node in a captured sequence is not safe on every HIP runtime this library
// meets (the 7.0 runtime PyTorch bundles replayed it with stale arguments once a second engine existed: profiles/r06x_graph_memset.log)
__global__ void __launch_bounds__(256) k_zero_u64(unsigned long long *__restrict__ p, unsigned n)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

// fixed-order sum of the per-wave partials of each job -> SUMS[slot][scale*18 + kind*3 + channel]; sums that no
// job produces (weight 0.0 in the reference's table) are written as 0
__global__ void __launch_bounds__(128) k_finish_jobs(TmJobs jobs, const double *__restrict__ PART, double *__restrict__ SUMS)
{
    const int i = threadIdx.x, slot = blockIdx.x;
    if (i >= 108) return;
    const int s = i / 18, kind = (i % 18) / 3, c = i % 3;
    const int j = jobs.job_of[s * 3 + c];
    double sum = 0.0;
    if (j >= 0 && (jobs.mode[j] == TM_MODE_FULL || (kind != 0 && kind != 3))) {
        // the partials are added in block order; their loads are issued eight at a time (a chain of dependent loads -- one memory
        // latency per row block, 17 of them for scale 0 of a 1080p frame -- was 8 of the 13 us this kernel took)
        const double *p = PART + ((size_t)slot * jobs.hstart[TM_MAX_JOBS]) * 6 + kind;
        const int b1 = jobs.hstart[j + 1];
        for (int b = jobs.hstart[j]; b < b1; b += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(b + k < b1 ? b + k : b1 - 1) * 6];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (b + k < b1) sum += v[k];
        }
    }
    SUMS[(size_t)slot * 108 + i] = sum;
}

} // namespace tmk
