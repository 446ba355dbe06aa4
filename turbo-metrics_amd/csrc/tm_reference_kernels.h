// tm_reference_kernels.h -- the straight-line, LDS-free kernels of the laboratory build (engine variant TM_VARIANT_REFERENCE):
//   k_ingest + k_downscale + k_xyb -> k_blur_v -> k_blur_h_jobs -> k_finish_jobs
// Their only job is to be obviously correct: the GPU tier checks that they and the tuned pipeline (tm_kernels.h) produce identical bits
// on the device, and both against the CPU oracle.  They keep the linear pyramid and a transposed XYB copy in HBM.  Not part of the
// ship build (`make ship`, -DTM_SHIP): included by tm_engine.hip and tests/emul only when the laboratory is wanted.
//
// Launch geometry (64-lane wavefront == 1 workgroup unless noted):
//   k_ingest          grid (ceil(ceil(w/2)/64), ceil(ceil(h/2)/4), slots)   block (64,4)
//   k_downscale       grid (ceil(dw/64), dh, slots*2*3)              block 64
//   k_xyb             grid (ceil(w/64), h, slots*2)                  block 64
//   k_blur_v          grid (vblk[6], 3, slots)                       block 64    lane = image column, all 6 scales in one launch
//   k_blur_h_jobs     grid (jobs.hstart[n], 1, slots)                block 64
#pragma once
#include "tm_kernels.h"

namespace tmk {

__global__ void __launch_bounds__(256) k_ingest(TmGeom g, const TmFrameDesc *__restrict__ desc,
                                                const float *__restrict__ lut, const float *__restrict__ coef,
                                                const double *__restrict__ tab, float *__restrict__ LIN,
                                                unsigned long long *__restrict__ SSE, int want_sse)
{
    const int qx = blockIdx.x * 64 + threadIdx.x;
    const int qy = blockIdx.y * 4 + threadIdx.y;
    const int slot = blockIdx.z;
    const int w = g.s[0].w, h = g.s[0].h, pitch = g.s[0].pitch;
    const bool inside = 2 * qx < w && 2 * qy < h;
    int q[2][2][2][3];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const TmFrameDesc d = desc[slot * 2 + side];
        float px[2][2][3];
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                for (int c = 0; c < 3; ++c) px[iy][ix][c] = 0.0f;
        if (inside) {
            if (d.kind == TM_KIND_NV12 || d.kind == TM_KIND_P016 || d.kind == TM_KIND_I420_8 || d.kind == TM_KIND_I420_16 || d.kind == TM_KIND_I420_P10) {
                if (2 * qx + 1 < w && 2 * qy + 1 < h) {
                    if (d.kind == TM_KIND_NV12 || d.kind == TM_KIND_I420_8) ingest_yuv_quad<unsigned char, 8>(d, coef, tab + TM_TAB_EOTF64, qx, qy, px);
                    else ingest_yuv_quad<unsigned short, 16>(d, coef, tab + TM_TAB_EOTF64, qx, qy, px);
                }
            } else {
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        const int x = 2 * qx + ix, y = 2 * qy + iy;
                        if (x < w && y < h) {
                            const char *row = (const char *)d.p0 + (size_t)y * d.pitch;
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                float v;
                                if (d.kind == TM_KIND_RGB8) v = lut[((const unsigned char *)row)[3 * x + c]];
                                else if (d.kind == TM_KIND_RGB16)
                                    v = tmdev::srgb_inverse_oetf((float)((const unsigned short *)row)[3 * x + c] / 65535.0f, tab);
                                else if (d.kind == TM_KIND_RGBF32)
                                    v = tmdev::srgb_inverse_oetf(((const float *)row)[3 * x + c], tab);
                                else v = ((const float *)row)[3 * x + c];
                                px[iy][ix][c] = v;
                            }
                        }
                    }
            }
            float *base = LIN + (size_t)(slot * 2 + side) * g.pyr;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int iy = 0; iy < 2; ++iy) {
                    const int y = 2 * qy + iy;
                    if (y < h) {
                        float *o = base + c * g.s[0].plane + (size_t)y * pitch + 2 * qx;
                        if (2 * qx + 1 < w) *(float2 *)o = make_float2(px[iy][0][c], px[iy][1][c]);
                        else o[0] = px[iy][0][c];
                    }
                }
        }
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix)
#pragma unroll
                for (int c = 0; c < 3; ++c) q[side][iy][ix][c] = (int)rintf(px[iy][ix][c] * 255.0f);
    }
    if (want_sse) {
        unsigned sse[3] = {0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                for (int ix = 0; ix < 2; ++ix) {
                    const int dlt = q[0][iy][ix][c] - q[1][iy][ix][c];
                    sse[c] += (unsigned)(dlt * dlt);
                }
        if (tm_wave_sum_u32x3(sse))
            for (int c = 0; c < 3; ++c) atomicAdd(&SSE[(size_t)slot * TM_SSE_BINS * 3 + c], (unsigned long long)sse[c]);
    }
}

// downscale_by_2, ssimulacra2-cuda-kernel/src/downscale.rs:5-35, one plane per blockIdx.z
__global__ void __launch_bounds__(64) k_downscale(TmGeom g, int s, float *__restrict__ LIN)
{
    const TmScaleGeom src = g.s[s - 1], dst = g.s[s];
    const int ox = blockIdx.x * 64 + threadIdx.x, oy = blockIdx.y;
    if (ox >= dst.w) return;
    const int img = blockIdx.z / 3, c = blockIdx.z % 3;
    const float *sp = LIN + (size_t)img * g.pyr + src.off + c * src.plane;
    float *dp = LIN + (size_t)img * g.pyr + dst.off + c * dst.plane;
    float sum = 0.0f;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const int x = min(ox * 2 + ix, src.w - 1);
            const int y = min(oy * 2 + iy, src.h - 1);
            sum += sp[(size_t)y * src.pitch + x];
        }
    dp[(size_t)oy * dst.pitch + ox] = sum * 0.25f;
}

// linear_to_xyb, ssimulacra2-cuda-kernel/src/xyb.rs:42-102 (planar in, planar out)
__global__ void __launch_bounds__(64) k_xyb(TmGeom g, int s, const float *__restrict__ LIN, float *__restrict__ XYB)
{
    const TmScaleGeom sg = g.s[s];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= sg.w) return;
    const size_t o = (size_t)blockIdx.z * g.pyr + sg.off + (size_t)y * sg.pitch + x;
    float X, Y, B;
    tmdev::linear_to_xyb(LIN[o], LIN[o + sg.plane], LIN[o + 2 * sg.plane], X, Y, B);
    XYB[o] = X;
    XYB[o + sg.plane] = Y;
    XYB[o + 2 * sg.plane] = B;
}

// ------------------------------------------------------------------------------------------------
// Column pass ("pass 1"): blur_plane_pass_fused down the columns of the five planes
// ref^2, dis^2, ref*dis, ref, dis  (ssimulacra2-cuda-kernel/src/blur.rs:34-137; which planes:
// ssimulacra2-cuda/src/lib.rs:299-335).  The three products (nppiMul, lib.rs:299-317) are formed in
// registers -- a rounded f32 multiply each, exactly what NPP stores -- so they never exist in HBM.
// One lane owns one column: 15 IIR sections (30 state registers) + a 20-row register window per
// input that is both the reference's 11-deep ring (blur.rs:25) and a 10-row load prefetch.
// Outputs are written TRANSPOSED (the reference's nppiTranspose, lib.rs:342-361,383-390): four
// consecutive rows of one column are 16 contiguous bytes there, so each lane stores one float4 per
// plane every four steps.  The row pass then reads everything row-contiguous.
// Step t reads row t (zero outside the image) and emits row t-4, t = 0 .. h+3 (blur.rs:95-136).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_blur_v(TmGeom g, const float *__restrict__ XYB, float *__restrict__ XYBT,
                                               float *__restrict__ V)
{
    int b = blockIdx.x, s = 0;
#pragma unroll
    for (int i = 1; i < TM_SCALES; ++i)
        if (b >= g.vblk[i]) s = i;
    const TmScaleGeom sg = g.s[s];
    const int x = (b - g.vblk[s]) * 64 + threadIdx.x;
    if (x >= sg.w) return;
    const int c = blockIdx.y, slot = blockIdx.z;
    const int h = sg.h, pitch = sg.pitch;
    const float *ref = XYB + (size_t)(slot * 2 + 0) * g.pyr + sg.off + c * sg.plane + x;
    const float *dis = XYB + (size_t)(slot * 2 + 1) * g.pyr + sg.off + c * sg.plane + x;
    const size_t to = sg.off_t + c * sg.plane_t + (size_t)x * sg.pitch_t;
    float *reft = XYBT + (size_t)(slot * 2 + 0) * g.pyr_t + to;
    float *dist = XYBT + (size_t)(slot * 2 + 1) * g.pyr_t + to;
    float *v0 = V + (size_t)(slot * 5 + 0) * g.pyr_t + to;
    float *v1 = V + (size_t)(slot * 5 + 1) * g.pyr_t + to;
    float *v2 = V + (size_t)(slot * 5 + 2) * g.pyr_t + to;
    float *v3 = V + (size_t)(slot * 5 + 3) * g.pyr_t + to;
    float *v4 = V + (size_t)(slot * 5 + 4) * g.pyr_t + to;

    float wr[20], wd[20];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        wr[j] = ld_row(ref, j, h, pitch);
        wd[j] = ld_row(dis, j, h, pitch);
        wr[j + 10] = 0.0f;
        wd[j + 10] = 0.0f;
    }
    tmdev::Iir f0 = {0, 0, 0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0, f4 = f0;
    float a0[4], a1[4], a2[4], a3[4], a4[4], ar[4], ad[4];
    const int T = h + 4;
    for (int t0 = 0; t0 < T; t0 += 20) {
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            const int t = t0 + j;
            const float r = wr[j], d = wd[j];
            const float rold = wr[(j + 10) % 20], dold = wd[(j + 10) % 20];
            wr[(j + 10) % 20] = ld_row(ref, t + 10, h, pitch);
            wd[(j + 10) % 20] = ld_row(dis, t + 10, h, pitch);
            a0[j & 3] = tmdev::iir_step(f0, rold * rold + r * r);
            a1[j & 3] = tmdev::iir_step(f1, dold * dold + d * d);
            a2[j & 3] = tmdev::iir_step(f2, rold * dold + r * d);
            a3[j & 3] = tmdev::iir_step(f3, rold + r);
            a4[j & 3] = tmdev::iir_step(f4, dold + d);
            ar[j & 3] = r;
            ad[j & 3] = d;
            if ((j & 3) == 3) {
                const int y0 = t - 7; // output rows y0..y0+3 (row t-4 is the newest)
                if (y0 >= 0 && y0 < h) {
                    *(float4 *)(v0 + y0) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                    *(float4 *)(v1 + y0) = make_float4(a1[0], a1[1], a1[2], a1[3]);
                    *(float4 *)(v2 + y0) = make_float4(a2[0], a2[1], a2[2], a2[3]);
                    *(float4 *)(v3 + y0) = make_float4(a3[0], a3[1], a3[2], a3[3]);
                    *(float4 *)(v4 + y0) = make_float4(a4[0], a4[1], a4[2], a4[3]);
                }
                const int r0 = t - 3; // input rows r0..r0+3, transposed copies for the edge terms
                if (r0 < h) {
                    *(float4 *)(reft + r0) = make_float4(ar[0], ar[1], ar[2], ar[3]);
                    *(float4 *)(dist + r0) = make_float4(ad[0], ad[1], ad[2], ad[3]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Row pass ("pass 2") fused with the error maps and the reductions: the reference's second
// blur_plane_pass_fused on the transposed images (lib.rs:368-379), compute_error_maps
// (error_maps.rs:5-60) and the six nppiSum / two nppiSqr per map (lib.rs:417-447) in one kernel.
// One lane owns one image ROW and walks x = 0..w-1 through the transposed planes (coalesced: lanes
// are consecutive y).  The blurred planes and the three maps never reach HBM; each lane keeps
// Sum(x) and Sum((x^2)^2) (squares rounded to f32, accumulation in f64, as NPP's Npp64f sums) and
// the wave total goes to PART; k_finish_jobs adds the partials of a job in a fixed order.
//
// Job driven: one wave = one 64-row block of one job.  FULL runs all five recurrences; EDGE runs only the
// mu1 / mu2 recurrences and the edge half of compute_error_maps, reading 4 planes instead of 7 (its freed
// registers go into a deeper load window: WN = 16 -> rows t+1 .. t+6 in flight).
// WN = window slots of the pass-1 planes (rows t-10 .. t+WN-11), WS = slots of the ref/dis windows
// (rows t-4 .. t+WN-11 need WN-6 slots; WS must divide WN).  PART[slot][row block over all jobs][6].
// grid (jobs.hstart[n], 1, slots), block 64.
// ------------------------------------------------------------------------------------------------
template <bool FULL, int WN, int WS>
__device__ __forceinline__ void blur_h_job(const float *__restrict__ reft, const float *__restrict__ dist,
                                           const float *__restrict__ v0, const float *__restrict__ v1,
                                           const float *__restrict__ v2, const float *__restrict__ v3,
                                           const float *__restrict__ v4, int w, int pt, bool valid, double (&acc)[6])
{
    static_assert(WN % WS == 0 && WS >= WN - 6, "window sizes");
    constexpr int P = WN - 10; // load distance in rows
    constexpr int NF = FULL ? WN : 1;
    float w0[NF], w1[NF], w2[NF], w3[WN], w4[WN], ws[WS], wq[WS];
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
#pragma unroll
    for (int j = 0; j < WS; ++j) {
        ws[j] = j < P ? ld_row(reft, j, w, pt) : 0.0f;
        wq[j] = j < P ? ld_row(dist, j, w, pt) : 0.0f;
    }
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
            const float src = ws[(j + WS - 4) % WS], dsv = wq[(j + WS - 4) % WS]; // row t-4
            if (FULL) {
                w0[(j + P) % NF] = ld_row(v0, t + P, w, pt);
                w1[(j + P) % NF] = ld_row(v1, t + P, w, pt);
                w2[(j + P) % NF] = ld_row(v2, t + P, w, pt);
            }
            w3[(j + P) % WN] = ld_row(v3, t + P, w, pt);
            w4[(j + P) % WN] = ld_row(v4, t + P, w, pt);
            ws[(j + P) % WS] = ld_row(reft, t + P, w, pt);
            wq[(j + P) % WS] = ld_row(dist, t + P, w, pt);
            if (t >= 4 && t < T) {
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

__global__ void __launch_bounds__(64) k_blur_h_jobs(TmGeom g, TmJobs jobs, const float *__restrict__ XYBT,
                                                    const float *__restrict__ V, double *__restrict__ PART)
{
    const int b = blockIdx.x;
    const int j = tm_find_job(jobs.hstart, b);
    const int s = jobs.scale[j], c = jobs.chan[j], mode = jobs.mode[j];
    const TmScaleGeom sg = g.s[s];
    const int y = (b - jobs.hstart[j]) * 64 + threadIdx.x;
    const bool valid = y < sg.h;
    const int yy = valid ? y : sg.h - 1;
    const int slot = blockIdx.z;
    const size_t to = sg.off_t + c * sg.plane_t + yy;
    const float *reft = XYBT + (size_t)(slot * 2 + 0) * g.pyr_t + to;
    const float *dist = XYBT + (size_t)(slot * 2 + 1) * g.pyr_t + to;
    const float *v0 = V + (size_t)(slot * 5 + 0) * g.pyr_t + to;
    const float *v1 = V + (size_t)(slot * 5 + 1) * g.pyr_t + to;
    const float *v2 = V + (size_t)(slot * 5 + 2) * g.pyr_t + to;
    const float *v3 = V + (size_t)(slot * 5 + 3) * g.pyr_t + to;
    const float *v4 = V + (size_t)(slot * 5 + 4) * g.pyr_t + to;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (mode == TM_MODE_FULL) blur_h_job<true, 12, 6>(reft, dist, v0, v1, v2, v3, v4, sg.w, sg.pitch_t, valid, acc);
    else blur_h_job<false, 16, 16>(reft, dist, v0, v1, v2, v3, v4, sg.w, sg.pitch_t, valid, acc);
    if (tm_wave_sum6(acc)) {
        double *o = PART + ((size_t)slot * jobs.hstart[TM_MAX_JOBS] + b) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = acc[k];
    }
}

} // namespace tmk
