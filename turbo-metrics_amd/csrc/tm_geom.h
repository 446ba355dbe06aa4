// tm_geom.h -- HBM layout of one engine (shared by host code and kernels; plain POD, passed to
// kernels by value in the kernarg segment so every field is wave-uniform / SGPR resident).
//
// All image data is PLANAR f32 (the reference keeps packed C3, which its own TODO at
// ssimulacra2-cuda/src/lib.rs:146-147 calls out as the coalescing problem).  Row pitches are
// multiples of 64 floats (256 B) so that every row starts on a cache-line pair and float4
// accesses at multiples of 4 are aligned.
//
//   "normal" orientation     : h rows of `pitch` floats, element (x,y) at y*pitch + x
//   "transposed" orientation : w rows of `pitch_t` floats, element (x,y) at x*pitch_t + y
//
// Arenas (float offsets):
//   LIN / XYB : [slot][side]   pyramid of 6 scales x 3 channel planes, normal
//   XYBT      : [slot][side]   same pyramid, transposed
//   V         : [slot][plane5] pyramid, transposed (output of the column pass: s11,s22,s12,mu1,mu2)
#pragma once
#include <stdint.h>

#define TM_SCALES 6
#define TM_SSE_BINS 64 /* integer SSE accumulators per slot (spreads the atomics of the ingest kernel) */

struct TmScaleGeom {
    int w, h;
    int pitch;   // floats
    int pitch_t; // floats
    unsigned long long plane;   // h * pitch
    unsigned long long plane_t; // round_up(w,64) * pitch_t
    unsigned long long off;     // float offset of this scale inside a normal pyramid
    unsigned long long off_t;   // float offset inside a transposed pyramid
};

struct TmGeom {
    TmScaleGeom s[TM_SCALES];
    unsigned long long pyr;   // floats per normal pyramid     (sum over scales of 3*plane)
    unsigned long long pyr_t; // floats per transposed pyramid (sum over scales of 3*plane_t)
    int vblk[TM_SCALES + 1];  // prefix sums: 64-column blocks of the column pass, per (slot, channel)
    int hblk[TM_SCALES + 1];  // prefix sums: 64-row blocks of the row pass, per (slot, channel)
};

// frame descriptor consumed by the ingest kernel
// I420_8 / I420_16: PLANAR 4:2:0 as files and software decoders deliver it (three planes; 16-bit samples little endian with
// the value in the LOW `16 - shift` bits): converted exactly like NV12 / P016 -- a 16-bit sample enters as v << shift, which is
// what an NVDEC P016 surface holds (cudarse-video/src/dec.rs:398-400)
// I420_P10: planar 4:2:0, 10-bit, THREE samples per 32-bit word (the upload kind for host frames: 10.7 instead of 16 bits per sample
// over PCIe).  A row of n samples is ceil(n / 384) blocks of 128 words; word k of block b holds samples 384 b + k (bits 0..9),
// 384 b + 128 + k (bits 10..19) and 384 b + 256 + k (bits 20..29), absent samples 0 (tm_p10_* below).  Three contiguous runs of 128
// samples per block: a host packs a row with plain vector code (three loads, two shifts, two ors), and a wave of the ingest kernel -- 128
// luma samples, 64 chroma samples -- finds all of its samples in ONE sub-run, behind a wave-uniform shift, as the same aligned 8-byte /
// 4-byte loads per lane the 16-bit kind uses.  The samples enter the conversion as v << 6, like I420_16 with shift 6: same bits.
enum { TM_KIND_NONE = -1, TM_KIND_NV12 = 0, TM_KIND_P016 = 1, TM_KIND_RGB8 = 2, TM_KIND_RGB16 = 3,
       TM_KIND_RGBF32 = 4, TM_KIND_LINEARF32 = 5, TM_KIND_I420_8 = 6, TM_KIND_I420_16 = 7, TM_KIND_I420_P10 = 8 };
#define TM_P10_RUN 128                     /* samples per contiguous run = words per block */
#define TM_P10_BLOCK (3 * TM_P10_RUN)      /* samples per block */
static inline unsigned long long tm_p10_row_words(unsigned long long n_samples) { return (n_samples + TM_P10_BLOCK - 1) / TM_P10_BLOCK * TM_P10_RUN; }

struct TmFrameDesc {
    const void *p0;            // luma plane or packed RGB
    const void *p1;            // interleaved CbCr plane (biplanar kinds) / Cb plane (planar kinds)
    const void *p2;            // Cr plane (planar kinds)
    unsigned long long pitch;  // bytes (luma / RGB rows; CbCr rows of the biplanar kinds)
    unsigned long long pitch2; // bytes, chroma rows of the planar kinds
    int kind;
    int matrix;
    int shift;                 // I420_16 / I420_P10: left shift that brings the sample to the top of 16 bits (6 for 10-bit content)
    int pad_;
};

// ---- job table of the two blur passes ---------------------------------------------------------------
// One job = one (scale, channel) image.  52 of the reference's 108 weights are non-zero
// (ssimulacra2-cuda/src/lib.rs:454-584), and a term with weight 0.0 contributes exactly 0.0 to the score
// (lib.rs:592-602 multiplies it away), so a job only has to produce the sums that carry weight:
//   TM_MODE_FULL  all five blurred planes (s11,s22,s12,mu1,mu2) -> ssim, artifact, detail_loss sums
//   TM_MODE_EDGE  mu1, mu2 only -> artifact, detail_loss sums (the ssim sums are reported as 0)
//   TM_MODE_NONE  nothing (every weight of this scale/channel is 0)
// With `full` set every job is TM_MODE_FULL: all 108 sums, exactly the reference's `scores` array.
#define TM_MAX_JOBS 18
enum { TM_MODE_NONE = 0, TM_MODE_EDGE = 1, TM_MODE_FULL = 2 };

struct TmJobs {
    int n;                        // jobs with mode != NONE
    int nfull;                    // jobs [0, nfull) run in the two blur passes; [nfull, n) are the EDGE jobs of the fused kernel
                                  // (k_blur_edge_fused); nfull == n when the table was made without `edge_last`
    int vstart[TM_MAX_JOBS + 1];  // prefix sums: workgroups of the column pass (FULL: one per 64-column block,
                                  // EDGE: one per two 64-column blocks)
    int hstart[TM_MAX_JOBS + 1];  // prefix sums: 64-row blocks (= waves) of the row pass; also indexes PART
    int scale[TM_MAX_JOBS], chan[TM_MAX_JOBS], mode[TM_MAX_JOBS];
    int job_of[TM_SCALES * 3];    // [scale*3 + channel] -> job index, -1 for TM_MODE_NONE
    int prio;                     // > 0: the two passes raise their waves' issue priority (they share the chip with the fused EDGE kernel)
};

static inline int tm_round_up(int v, int m) { return (v + m - 1) / m * m; }

static inline void tm_make_geom(TmGeom *g, int w, int h)
{
    unsigned long long off = 0, off_t = 0;
    g->vblk[0] = 0;
    g->hblk[0] = 0;
    for (int i = 0; i < TM_SCALES; ++i) {
        TmScaleGeom *s = &g->s[i];
        s->w = w; s->h = h;
        s->pitch = tm_round_up(w, 64);
        s->pitch_t = tm_round_up(h, 64);
        s->plane = (unsigned long long)h * s->pitch;
        s->plane_t = (unsigned long long)tm_round_up(w, 64) * s->pitch_t; // rows padded: see blur_v_flush
        s->off = off; s->off_t = off_t;
        off += 3 * s->plane; off_t += 3 * s->plane_t;
        g->vblk[i + 1] = g->vblk[i] + (w + 63) / 64;
        g->hblk[i + 1] = g->hblk[i] + (h + 63) / 64;
        // scale sizes: ssimulacra2-cuda/src/lib.rs:62-66
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    g->pyr = off; g->pyr_t = off_t;
}

// weights: the reference's table [channel][scale][ssim1, art1, det1, ssim4, art4, det4]
// edge_last: the EDGE jobs come after ALL FULL jobs (they run in k_blur_edge_fused; the two passes then launch only the
// workgroups of jobs [0, nfull), and PART keeps one layout for k_finish_jobs)
static inline void tm_make_jobs(TmJobs *j, const TmGeom *g, const double *weights, int full, int edge_last = 0)
{
    j->n = 0; j->nfull = 0; j->prio = 0;
    j->vstart[0] = 0; j->hstart[0] = 0;
    for (int i = 0; i < TM_SCALES * 3; ++i) j->job_of[i] = -1;
    // longest columns/rows first (scale 0), FULL before EDGE inside a scale
    for (int round = 0; round < (edge_last ? 2 : 1); ++round)
    for (int s = 0; s < TM_SCALES; ++s)
        for (int pass = TM_MODE_FULL; pass >= TM_MODE_EDGE; --pass)
            for (int c = 0; c < 3; ++c) {
                if (edge_last && pass != (round == 0 ? TM_MODE_FULL : TM_MODE_EDGE)) continue;
                const double *wt = weights + c * 36 + 6 * s;
                const int ssim = wt[0] != 0.0 || wt[3] != 0.0;
                const int edge = wt[1] != 0.0 || wt[2] != 0.0 || wt[4] != 0.0 || wt[5] != 0.0;
                const int mode = full || ssim ? TM_MODE_FULL : (edge ? TM_MODE_EDGE : TM_MODE_NONE);
                if (mode != pass) continue;
                const int k = j->n++;
                const int cb = (g->s[s].w + 63) / 64, rb = (g->s[s].h + 63) / 64;
                j->scale[k] = s; j->chan[k] = c; j->mode[k] = mode;
                j->vstart[k + 1] = j->vstart[k] + (mode == TM_MODE_FULL ? cb : (cb + 1) / 2);
                j->hstart[k + 1] = j->hstart[k] + rb;
                j->job_of[s * 3 + c] = k;
                if (!edge_last || mode == TM_MODE_FULL) j->nfull = j->n;
            }
    for (int k = j->n; k < TM_MAX_JOBS; ++k) {
        j->scale[k] = 0; j->chan[k] = 0; j->mode[k] = TM_MODE_NONE;
        j->vstart[k + 1] = j->vstart[k]; j->hstart[k + 1] = j->hstart[k];
    }
}
