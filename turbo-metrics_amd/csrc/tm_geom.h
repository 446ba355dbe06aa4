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
enum { TM_KIND_NONE = -1, TM_KIND_NV12 = 0, TM_KIND_P016 = 1, TM_KIND_RGB8 = 2, TM_KIND_RGB16 = 3,
       TM_KIND_RGBF32 = 4, TM_KIND_LINEARF32 = 5 };

struct TmFrameDesc {
    const void *p0;           // luma plane or packed RGB
    const void *p1;           // interleaved CbCr plane (YUV kinds)
    unsigned long long pitch; // bytes
    int kind;
    int matrix;
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
