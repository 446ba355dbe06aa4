/*
 * turbo_metrics_hip.h -- C ABI of the MI355X (gfx950) SSIMULACRA2 / PSNR frame-pair engine.
 *
 * This is the drop-in boundary for the hot path of Gui-Yom/turbo-metrics: everything the
 * reference does between "a decoded frame pair is in device memory" and "FrameScores".
 * It replaces, as one fused engine, the reference's Rust->cudarse FFI for this path:
 *
 *   reference interface (file:line, relative to crates/)               entry point here
 *   -----------------------------------------------------------------  -------------------------------
 *   turbo-metrics/src/lib.rs:438-456  init_cuda()                      tm_init
 *   turbo-metrics/src/lib.rs:201-249  TurboMetrics::new(w,h,&Metrics)  tm_engine_create
 *   ssimulacra2-cuda/src/lib.rs:48-107 Ssimulacra2::new (buffers)      tm_engine_create
 *   ssimulacra2-cuda/src/lib.rs:110   Ssimulacra2::mem_usage           tm_engine_mem_usage
 *   turbo-metrics/src/color.rs:96-116 convert_frame_to_linearrgb       tm_engine_set_frame_{nv12,p016,
 *     + cuda-colorspace/src/lib.rs:33-170 ColorspaceConversion::*        rgb8,rgb16,rgbf32} (+ _i420: planar)
 *   ssimulacra2-cuda/src/lib.rs:48-52 (linear f32 C3 inputs of
 *     Ssimulacra2::new / compute)                                      tm_engine_set_frame_linear_f32
 *   turbo-metrics/src/lib.rs:268-345  compute_one (launch part)        tm_engine_compute_async
 *   ssimulacra2-cuda/src/lib.rs:283-286 Ssimulacra2::compute           tm_engine_compute_async
 *   turbo-metrics/src/lib.rs:347-359  compute_one (sync + FrameScores) tm_engine_sync, tm_engine_get_scores
 *   ssimulacra2-cuda/src/lib.rs:271-291 compute_sync / get_score       tm_engine_sync, tm_engine_get_scores
 *   ssimulacra2-cuda/src/lib.rs:44 `scores: [f64;108]` (pre post-proc) tm_engine_get_raw_sums
 *   cudarse-driver-sys/src/lib.rs:12-31 CuError Display                tm_strerror
 *
 * Differences from the reference that are part of the contract:
 *   - One engine holds `batch_capacity` frame-pair SLOTS.  set_frame fills (slot, side);
 *     compute_async runs the whole pipeline for slots [0, n_slots) in a handful of launches.
 *     batch_capacity = 1 reproduces the reference's one-pair-at-a-time `compute_one`.
 *   - No panics / exceptions cross the ABI: every call returns a TM_* code.  The combinations the
 *     reference leaves as `todo!()` (full-range YUV, non-BT.709 transfer:
 *     cuda-colorspace/src/lib.rs:45-52) return TM_ERR_UNSUPPORTED.
 *   - One engine = one device = one host thread at a time (same as the reference, lib.rs:438-456).
 *
 * Plain C types only, so a Rust `extern "C"` block, cgo, JNI or ctypes can bind it unchanged
 * (INTEGRATION.md shows the Rust binding).
 *
 * NUMERICAL CONTRACT.  Integer results (the PSNR sum of squared errors) are exact.  Every f32 plane the kernels produce is
 * bit-identical to the CPU oracle of this build (oracle/tm_oracle.c; a second, independently written numpy restatement agrees
 * with it bit for bit), so scores agree with that oracle to <= 1e-9.  Against the reference's own binary bit equality cannot
 * be claimed: it calls closed NVIDIA code (__nv_fast_powf, ~8 ulp; __nv_cbrtf, 1 ulp) whose last bits the score amplifies.  This
 * build evaluates the reference's expressions more accurately than the reference does -- the BT.709 transfer function is the
 * reference's expression correctly rounded (its own f32 base (v + a) / A, a binary64 cubic, one rounding: 117 of 15.4 M arguments
 * are not the nearest float); the cube root is within 0.5003 ulp -- and tests/golden/scores_accurate_frozen.json commits, per
 * seeded case, the distance to the correctly rounded evaluation of every expression: 3e-6 ... 2.4e-3 (all of it the cube root's;
 * an exp2f(y * log2f(x))-shaped pow like libdevice's moves these cases by 3e-4 ... 1.1e-2, and the 1080p NV12 case moves by
 * 7e-3 ... 4e-2 when 0.8 % of its linear samples move by ONE ulp; the reference's own GPU-vs-CPU check allows 0.25).  Each case is
 * held to max(2 x its committed distance, 1e-3) by tests/test_golden_accurate.py.
 */
#ifndef TURBO_METRICS_HIP_H
#define TURBO_METRICS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tm_engine tm_engine;

/* return codes */
enum {
    TM_OK = 0,
    TM_ERR_INVALID_ARG = 1,
    TM_ERR_UNSUPPORTED = 2, /* reference `todo!()` territory */
    TM_ERR_HIP = 3,         /* a HIP runtime call failed; see tm_last_hip_error() */
    TM_ERR_OOM = 4,
    TM_ERR_STATE = 5        /* e.g. get_scores before compute, slot not filled */
};

/* metrics_mask bits == the reference's `Metrics` struct (turbo-metrics/src/lib.rs:27-37) */
enum { TM_METRIC_PSNR = 1, TM_METRIC_SSIM = 2, TM_METRIC_MSSSIM = 4, TM_METRIC_SSIMULACRA2 = 8 };

/* cuda_colorspace::ColorMatrix / Transfer (cuda-colorspace/src/lib.rs) */
enum { TM_MATRIX_BT709 = 0, TM_MATRIX_BT601_525 = 1, TM_MATRIX_BT601_625 = 2 };
enum { TM_TRANSFER_BT709 = 0 };

enum { TM_SIDE_REF = 0, TM_SIDE_DIS = 1 };

/* where the frame bytes live.  TM_MEM_DEVICE is zero-copy: the pointers are read by the
 * ingest kernel at compute time and must stay valid until tm_engine_sync returns (the
 * reference has the same rule for mapped NVDEC surfaces). TM_MEM_HOST is copied into an
 * engine-owned device surface on the engine's stream before the call returns control. */
enum { TM_MEM_HOST = 0, TM_MEM_DEVICE = 1, TM_MEM_HOST_PINNED = 2 };
/* TM_MEM_HOST_PINNED: page-locked host memory (tm_host_alloc): copied into the engine-owned surface by an asynchronous
 * DMA on the engine's stream; like TM_MEM_DEVICE the bytes must stay untouched until tm_engine_sync returns. */

/* == turbo_metrics::FrameScores (lib.rs:112-123); `valid` has the TM_METRIC_* bits of the
 * Option<> fields that are Some(). */
typedef struct tm_frame_scores {
    double psnr;
    double ssim;
    double msssim;
    double ssimulacra2;
    uint32_t valid;
} tm_frame_scores;

/* Bind the calling process to `device` (hipSetDevice) and make sure a gfx950 device is there.
 * Fails loudly (TM_ERR_HIP / TM_ERR_UNSUPPORTED) when no usable GPU exists: there is no CPU path. */
int tm_init(int device);
/* number of visible devices (0 when the runtime finds none); does not bind the process to any */
int tm_device_count(void);

/* page-locked host memory for frame staging (hipHostMalloc / hipHostFree); NULL when out of memory */
void *tm_host_alloc(size_t bytes);
void tm_host_free(void *p);

/* Placement search at engine creation (process-wide setting, default 8, 1 = off; also TM_PLACEMENT_CANDIDATES in the environment).
 * The column pass stores transposed 128-B lines over the whole pass-1 arena, and how fast that goes depends on the PHYSICAL
 * backing of the allocation: the same kernel takes 2.14 ... 2.47 ms per 64 1080p pairs from one allocation to the next, while
 * moving the arena's start inside one allocation changes nothing (profiles/r02d_v_offset_probe.json) -- so there is no
 * alignment rule to apply instead.  For arenas of 1 GiB and more tm_engine_create therefore allocates up to `n` candidates,
 * brings the device to its steady clock (~120 ms of the two kernels: a cold device runs 10-15 % slower, more than the placements
 * differ), times the column pass and the row pass on every candidate in turns (neither has a data-dependent branch) and keeps
 * the fastest; the others are freed before it returns.  Costs ~0.2 s and, WHILE IT RUNS, up to n times the arena's memory (not reflected by
 * tm_engine_mem_usage afterwards): it stops early when the candidates would hold more than a third of the device's memory or
 * when allocating the next one would leave less than a quarter of the device free. */
void tm_set_placement_candidates(int n);

int tm_engine_create(tm_engine **out, uint32_t width, uint32_t height, uint32_t metrics_mask,
                     uint32_t batch_capacity);
void tm_engine_destroy(tm_engine *e);

/* bytes of device memory held by the engine */
size_t tm_engine_mem_usage(const tm_engine *e);

/* Frame surfaces: `pitch` (bytes between rows) must be at least one row, below 2^24, and pitch * rows below 4 GB -- the ingest
 * kernel addresses a surface with 32-bit lane offsets; anything else is TM_ERR_INVALID_ARG.
 *
 * NV12: 8-bit luma plane `y` (rows at `pitch` bytes) + interleaved CbCr plane `uv` (same pitch);
 * layout of an NVDEC mapping, cudarse-video/src/dec.rs:299-346.  Visible window starts at the origin.
 * Host surfaces (TM_MEM_HOST, TM_MEM_HOST_PINNED): when `uv` lies a whole number of rows R behind `y`, height <= R <= height + 64,
 * the two planes are taken as ONE allocation -- the reference's decoded-surface contract (one allocation, luma rows of the coded
 * height, then the CbCr rows) -- and go up in one 2-D copy that also reads the R - height padding rows in between.  Planes that
 * live in separate allocations must therefore not sit at such a distance from each other by accident; any other distance gets
 * one copy per plane. */
int tm_engine_set_frame_nv12(tm_engine *e, uint32_t slot, int side, const void *y, const void *uv,
                             size_t pitch, int matrix, int transfer, int full_range, int mem);
/* P016: 16-bit samples, 10-bit values MSB aligned (dec.rs:348-403). `pitch` in bytes.
 * For TM_MEM_HOST the engine copies `height` luma rows and ceil(height/2) chroma rows. */
int tm_engine_set_frame_p016(tm_engine *e, uint32_t slot, int side, const void *y, const void *uv,
                             size_t pitch, int matrix, int transfer, int full_range, int mem);
/* Planar 4:2:0 as files and software decoders deliver it (I420 / yuv420p: bits = 8; yuv420p10le etc.: bits = 9..16, little
 * endian u16 with the value in the LOW bits): luma plane `y` at pitch_y bytes, chroma planes `u` (Cb) and `v` (Cr) of
 * ceil(w/2) x ceil(h/2) samples at pitch_uv bytes.  Not a reference format: its decoder only ever hands over NV12 / P016
 * (cudarse-video/src/dec.rs:299-403), so a host that holds planar pictures would have to repack them first -- this entry point
 * lets it upload the file bytes as they are.  The arithmetic is the NV12 / P016 conversion of the same samples: a 16-bit
 * sample v enters as v << (16 - bits), exactly what the repacked P016 surface would hold (dec.rs:398-400); results are
 * bit-identical to tm_engine_set_frame_{nv12,p016} on the repacked surface (tests/test_gpu_parity.py). */
int tm_engine_set_frame_i420(tm_engine *e, uint32_t slot, int side, const void *y, const void *u, const void *v,
                             size_t pitch_y, size_t pitch_uv, int bits, int matrix, int transfer, int full_range, int mem);
/* packed RGB, 3 samples per pixel (HwFrame::Npp8 / Npp16 / Npp32, turbo-metrics/src/lib.rs:125-130);
 * sRGB transfer: 8-bit through the LUT, 16-bit / f32 through the formula (color.rs:112-114). */
int tm_engine_set_frame_rgb8(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem);
int tm_engine_set_frame_rgb16(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem);
int tm_engine_set_frame_rgbf32(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem);
/* already-linear packed RGB f32: the input of ssimulacra2_cuda::Ssimulacra2 itself */
int tm_engine_set_frame_linear_f32(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem);

/* Enqueue the whole pipeline for slots [0, n_slots) on the engine's stream; returns immediately. */
int tm_engine_compute_async(tm_engine *e, uint32_t n_slots);
/* Block until the last compute_async has finished and its results are on the host. */
int tm_engine_sync(tm_engine *e);
/* FrameScores of one slot of the last completed compute (host-side post-processing happens here) */
int tm_engine_get_scores(tm_engine *e, uint32_t slot, tm_frame_scores *out);
/* the same for slots [first_slot, first_slot + n) in one call: out[0 .. n) (what compute_all collects per batch) */
int tm_engine_get_scores_batch(tm_engine *e, uint32_t first_slot, uint32_t n, tm_frame_scores *out);
/* the 108 raw sums [scale][kind][channel] == the reference's `scores` before post_process_scores
 * (ssimulacra2-cuda/src/lib.rs:417-447).  By default the 56 sums whose weight in the reference's table is
 * exactly 0.0 (lib.rs:454-584; they are multiplied away at lib.rs:592-602) are not computed and read 0.0;
 * the score is bit-identical either way.  tm_engine_set_full_sums(e, 1) computes all 108. */
int tm_engine_get_raw_sums(tm_engine *e, uint32_t slot, double out[108]);
int tm_engine_set_full_sums(tm_engine *e, int on);
/* what the blur passes compute per [scale*3 + channel]: 0 nothing, 1 edge terms only (mu1, mu2), 2 everything */
int tm_engine_get_job_modes(const tm_engine *e, int out[18]);
/* PSNR input: exact integer sum of squared differences of the u8-quantised linear RGB pair (all three channels; and per
 * channel R, G, B) */
int tm_engine_get_sse(tm_engine *e, uint32_t slot, uint64_t *out);
int tm_engine_get_sse_channels(tm_engine *e, uint32_t slot, uint64_t out[3]);
/* 10 log10(255^2 n / sse) as the single Npp32f the reference reads back, widened (turbo-metrics/src/lib.rs:355) */
double tm_psnr_from_sse(uint64_t sse, uint64_t n_samples);

/* How PSNR / SSIM / MS-SSIM combine the three channels (NPP's nppi*_8u_C3R is closed source; the reference reads back one
 * float, cudarse-npp/src/image/ist.rs:118-133):
 *   TM_CHANNELS_POOLED (default)  PSNR from the MSE over all 3*w*h samples; SSIM / MS-SSIM = mean of the three channels
 *   TM_CHANNELS_FIRST             the value of channel 0 (R) only -- what the reference would see if NPP writes one value
 *                                 per channel into the buffer of which it reads the first */
enum { TM_CHANNELS_POOLED = 0, TM_CHANNELS_FIRST = 1 };
int tm_engine_set_channel_mode(tm_engine *e, int mode);

/* SSIM / MS-SSIM of the u8-quantised linear RGB pair (the inputs of nppiSSIM_8u_C3R_Ctx / nppiWMSSSIM_8u_C3R_Ctx,
 * turbo-metrics/src/lib.rs:319-337).  NPP is closed source and no reference test pins its output: the definition here
 * is BUILD-DEFINED (Wang et al. 2004 / 2003) -- 11x11 Gaussian window (sigma 1.5) over the windows
 * inside the image, K1 = 0.01, K2 = 0.03, L = 255, per channel then averaged; MS-SSIM: five dyadic scales.
 * Raw sums [channel 3][scale 5][sum of ssim, sum of cs] over the (w-10) x (h-10) windows of each scale (SSIM alone fills
 * scale 0 only); the two host functions turn them into the scores that tm_engine_get_scores reports. */
/* By default the sum of l * cs is only produced where a score reads it (scale 0 for SSIM, scale 4 for MS-SSIM, which uses the
 * contrast-structure term alone on scales 0..3) and reads 0.0 elsewhere; tm_engine_set_full_sums(e, 1) produces every entry
 * of the scales that were run. */
int tm_engine_get_ssim_sums(tm_engine *e, uint32_t slot, double out[30]);
double tm_ssim_from_sums(const double sums[30], uint32_t width, uint32_t height);
double tm_msssim_from_sums(const double sums[30], uint32_t width, uint32_t height);
double tm_ssim_channel_from_sums(const double sums[30], uint32_t width, uint32_t height, int channel);
double tm_msssim_channel_from_sums(const double sums[30], uint32_t width, uint32_t height, int channel);
void tm_ssim_window(float g[11]);

/* SSIMULACRA2 post-processing (ssimulacra2-cuda/src/lib.rs:449-623) as a pure host function */
double tm_ssimulacra2_score_from_sums(const double sums[108], uint32_t width, uint32_t height);

/* ---- measurement hooks -------------------------------------------------------------- */
enum { TM_STAGE_INGEST = 0, TM_STAGE_BLUR_V = 1, TM_STAGE_BLUR_H = 2, TM_STAGE_SSIM = 3 /* sum finisher + SSIM / MS-SSIM kernels */, TM_STAGE_COUNT = 4 };
/* When on, HIP events bracket each stage of every compute_async on the engine's own stream. */
int tm_engine_set_profiling(tm_engine *e, int on);
/* Accumulated since the last reset: milliseconds per stage and number of computes measured. */
int tm_engine_get_stage_ms(tm_engine *e, double ms[TM_STAGE_COUNT], uint64_t *n_computes, int reset);
/* 1: the per-batch sequence (descriptor upload, kernels, result download) is captured once into a hipGraph and replayed --
 * the counterpart of the reference's recorded CUDA graph (ssimulacra2-cuda/src/lib.rs:140-229).  0 (default): direct
 * launches; with ~10 submissions per batch instead of the reference's 305 the graph has nothing left to hide and measured
 * 1-6 % slower on ROCm 7.2. */
int tm_engine_set_graph(tm_engine *e, int on);
/* Which kernels run.  TM_VARIANT_DEFAULT: the tuned pipeline (csrc/tm_kernels.h).  TM_VARIANT_REFERENCE: the straight-line,
 * LDS-free kernels kept as the on-device cross-check (SSIMULACRA2 / PSNR only; they keep the linear pyramid and a transposed
 * XYB copy in HBM, allocated on first selection) -- the two produce identical bits (tests/test_gpu_parity.py).
 * TM_VARIANT_WIDE_ROWS (test hook, default pipeline only): the row-pass instantiation that frames wider than 2560 pixels get,
 * forced on any size.  TM_VARIANT_TILE_INGEST: the 32 x 8 tile ingest kernel for the 4:2:0 kinds too (default: the row-walking
 * kernel).  TM_VARIANT_SPLIT_ROWS / TM_VARIANT_WHOLE_ROWS: force / forbid the three-wave row pass that small batches get by
 * default.  Every combination produces the same bits.  TM_ERR_INVALID_ARG for any other value. */
enum { TM_VARIANT_DEFAULT = 0, TM_VARIANT_REFERENCE = 1, TM_VARIANT_WIDE_ROWS = 0x100, TM_VARIANT_TILE_INGEST = 0x200, TM_VARIANT_SPLIT_ROWS = 0x400,
       TM_VARIANT_WHOLE_ROWS = 0x800 };
int tm_engine_set_variant(tm_engine *e, int variant);

/* ---- test hooks: read back intermediate planes of one slot (blocking) ------------------ */
enum {
    TM_PLANE_LINEAR = 0, /* index = side,            channel = R,G,B  ; scale 0..5 ; w x h   (reference pipeline only) */
    TM_PLANE_XYB = 1,    /* index = side,            channel = X,Y,B  ; w x h                 */
    TM_PLANE_XYB_T = 2,  /* index = side,            transposed: h wide, w tall               (reference pipeline only) */
    TM_PLANE_PASS1_T = 3 /* index = 0..4 (s11,s22,s12,mu1,mu2), transposed: h wide, w tall    */
};
int tm_engine_debug_read_plane(tm_engine *e, uint32_t slot, int kind, int scale, int index, int channel,
                               float *out, size_t out_count);

/* measurement hook: move the start of the pass-1 arena by `bytes` (multiple of 16, <= 4 MiB) inside its allocation -- how the
 * column pass reacts to the arena's alignment can then be measured on ONE allocation (tools/v_offset_probe.py) */
int tm_engine_debug_set_v_offset(tm_engine *e, size_t bytes);
/* measurement hook: quad rows one wave of the 4:2:0 ingest kernel (k_ingest_rows) walks; even, 2..128; 0 = chosen per launch
 * (the default; the environment variable TM_INGEST_ROWS sets it at creation) -- results do not depend on it (tools/ingest_ab.py) */
int tm_engine_debug_set_ingest_rows(tm_engine *e, int rows);

const char *tm_strerror(int code);
/* text of the last HIP error seen by this thread's calls ("" if none) */
const char *tm_last_hip_error(void);
const char *tm_version(void);

#ifdef __cplusplus
}
#endif
#endif
