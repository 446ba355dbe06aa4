/*
 * turbo_metrics_hip_debug.h -- the laboratory side of libturbometrics_hip.so: kernel variants, tuning values, fault injection,
 * stage timers and plane read-back.  Exported by the same library as include/turbo_metrics_hip.h, used by tests/, tools/ and
 * bench.py's roofline measurement; a caller that binds the engine never needs any of it, and no result ever
 * depends on a tuning value.  The reference has no counterpart (its kernels have one configuration).
 */
#ifndef TURBO_METRICS_HIP_DEBUG_H
#define TURBO_METRICS_HIP_DEBUG_H

#include "turbo_metrics_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostics on stderr (process-wide, default off): what the placement search measured per candidate */
void tm_set_debug_log(int on);
/* what the blur passes compute per [scale*3 + channel]: 0 nothing, 1 edge terms only (mu1, mu2), 2 everything */
int tm_engine_get_job_modes(const tm_engine *e, int out[18]);
/* 1 when a compute_async of n_slots slots runs its edge-only jobs in the fused kernel (k_blur_edge_fused: see TM_VARIANT_*), 0 when
 * the two blur passes run them; < 0 on a bad argument */
int tm_engine_uses_fused_edge(const tm_engine *e, uint32_t n_slots);
/* per-channel forms of tm_ssim_from_sums / tm_msssim_from_sums (what TM_CHANNELS_FIRST reports for channel 0) and the 11 taps of
 * the normalised Gaussian window the SSIM kernels use */
double tm_ssim_channel_from_sums(const double sums[30], uint32_t width, uint32_t height, int channel);
double tm_msssim_channel_from_sums(const double sums[30], uint32_t width, uint32_t height, int channel);
void tm_ssim_window(float g[11]);

/* ---- measurement hooks -------------------------------------------------------------- */
enum { TM_STAGE_INGEST = 0, TM_STAGE_BLUR_V = 1, TM_STAGE_BLUR_H = 2, TM_STAGE_SSIM = 3 /* sum finisher + SSIM / MS-SSIM kernels (+ the wait for the fused kernel, should it still run) */,
       TM_STAGE_EDGE = 4 /* k_blur_edge_fused + k_finish_edge: the edge-only jobs in one kernel.  By default it runs on a second stream BESIDE
                          * BLUR_V / BLUR_H (its own pair of events): the stage times then overlap and do not add up to the step */, TM_STAGE_COUNT = 5 };
/* When on, HIP events bracket each stage of every compute_async on the engine's own stream. */
int tm_engine_set_profiling(tm_engine *e, int on);
/* Accumulated since the last reset: milliseconds per stage and number of computes measured. */
int tm_engine_get_stage_ms(tm_engine *e, double ms[TM_STAGE_COUNT], uint64_t *n_computes, int reset);
/* Which kernels run.  TM_VARIANT_DEFAULT: the tuned pipeline (csrc/tm_kernels.h).  TM_VARIANT_REFERENCE: the straight-line,
 * LDS-free kernels kept as the on-device cross-check (SSIMULACRA2 / PSNR only; they keep the linear pyramid and a transposed
 * XYB copy in HBM, allocated on first selection) -- the two produce identical bits (tests/test_gpu_parity.py).
 * TM_VARIANT_WIDE_ROWS (test hook, default pipeline only): the row-pass instantiation that frames wider than 2560 pixels get,
 * forced on any size.  TM_VARIANT_TILE_INGEST: the 32 x 8 tile ingest kernel for the 4:2:0 kinds too (default: the row-walking
 * kernel).  TM_VARIANT_SPLIT_ROWS / TM_VARIANT_WHOLE_ROWS: force / forbid the eight-wave row pass that small launches get by
 * default.  TM_VARIANT_TWO_PASS_EDGE / TM_VARIANT_FUSED_EDGE: forbid / force the fused kernel for the EDGE jobs (the planes of
 * which only the two edge maps carry weight: scale 0 of X and B): by default launches with 400 and more bands of 32 rows of such planes (6 pairs of 1080p, 3 of 4K) run both
 * recurrences, the maps and the sums of those jobs in one kernel (k_blur_edge_fused) that never writes their pass-1 planes --
 * the TM_PLANE_PASS1_T read-back of such a job then returns what an earlier launch left there.
 * Every combination produces the same bits.  TM_ERR_INVALID_ARG for any other value. */
enum { TM_VARIANT_DEFAULT = 0, TM_VARIANT_REFERENCE = 1, TM_VARIANT_WIDE_ROWS = 0x100, TM_VARIANT_TILE_INGEST = 0x200, TM_VARIANT_SPLIT_ROWS = 0x400,
       TM_VARIANT_WHOLE_ROWS = 0x800, TM_VARIANT_TWO_PASS_EDGE = 0x1000, TM_VARIANT_UPPER_KERNEL = 0x2000, TM_VARIANT_FUSED_EDGE = 0x4000 };
/* TM_VARIANT_UPPER_KERNEL: pyramid levels 2..5 of the 4:2:0 kinds by the second kernel (k_ingest_upper_rd, through the LIN2 arena: the
 * arrangement up to round 5, and still that of the tile kernel) instead of the row-walking kernel's own epilogue. */
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
 * (the default) -- results do not depend on it (tools/ingest_ab.py) */
int tm_engine_debug_set_ingest_rows(tm_engine *e, int rows);
/* measurement hook: where the fused kernel of the edge-only jobs runs.  1 (default): on the engine's second stream BESIDE the two
 * blur passes, enqueued before the column pass; 2: the same, enqueued after it; 0: behind the row pass on the engine's stream,
 * every kernel alone on the chip (per-kernel timings that mean one kernel; the step is 4-15 % slower).  Results do not depend on
 * it.  The second stream is one per device and process, shared by the
 * engines on that device (the runtime has few hardware queues: a stream more per engine slows the uploads of a ping-pong pair);
 * with the hipGraph replay on (tm_engine_set_graph) the kernel stays on the engine's own stream. */
int tm_engine_debug_set_edge_beside(tm_engine *e, int mode);
/* test hook: the launch epoch of the fused kernel's hand-off tags (24 bits, never 0; it advances by one per launch and wraps to 1;
 * the launch that finds it at 1 clears the hand-off words first, so that no tag of 2^24 launches ago can match) */
int tm_engine_debug_set_edge_epoch(tm_engine *e, uint32_t epoch);
/* Tuning values and fault injection, per engine (tools/ and tests; a release caller never needs them -- no environment variable
 * reaches any of these).  Results never depend on the tuning values.
 *   TM_DBG_FUSED_EDGE_FROM     bands of 32 rows of edge-only planes per launch from which those jobs take the fused kernel (400)
 *   TM_DBG_EF_WAVES            fused kernel: 4 = four adjacent bands of one plane per workgroup (default), 5 = the same band of four
 *                              planes, 1 = single-wave workgroups
 *   TM_DBG_EF_PERSIST_WGS      its workgroups when it runs beside the passes: 0 = 7/8 per CU (default), > 0 = that many, -1 = one per ticket
 *   TM_DBG_PASS_PRIO           1 (default): the two passes raise their waves' priority while the fused kernel runs beside them
 *   TM_DBG_SPLIT_ROWS_BELOW    row blocks per launch up to which the eight-wave row pass runs (1 024; 2 600 beside the fused kernel)
 *   TM_DBG_SOLO_COL_BELOW      role-waves of the column pass per launch (five per column block) up to which each runs as a workgroup of its own
 *   TM_DBG_LINEAR_UPLOAD       1: a tight planar picture in host memory (tm_engine_set_frame_i420: no row padding, Cb behind Y, Cr
 *                              behind Cb) goes up as one linear copy; 0 (default): as 2-D copies into padded rows like every other layout
 *   TM_DBG_UPLOAD_STREAMS      2 (default): page-locked frames of the distorted side go up on a second stream (one per device, shared by its
 *                              engines) beside those of the reference side; 1: every frame on the engine's own stream
 *   TM_DBG_UPLOAD_MERGE        bytes up to which two page-locked frames that lie back to back in the caller's memory (and therefore in the
 *                              engine's staging arena) go up as ONE DMA (14 MiB: up to four 1080p frames; 0: every frame its own copy, at once)
 *   TM_DBG_EF_FAULT            fault injection: 2 = the fused kernel does not publish the column state between groups of bands -- the next
 *                              group's wait times out, tm_engine_sync returns TM_ERR_HIP and the results of that launch are not
 *                              available; 1 = do not wait at all (wrong sums, no error); 0 = off.  The engine stays usable. */
enum { TM_DBG_FUSED_EDGE_FROM = 0, TM_DBG_EF_WAVES = 1, TM_DBG_EF_PERSIST_WGS = 2, TM_DBG_PASS_PRIO = 3, TM_DBG_SPLIT_ROWS_BELOW = 4, TM_DBG_SOLO_COL_BELOW = 5, TM_DBG_EF_FAULT = 6, TM_DBG_LINEAR_UPLOAD = 7, TM_DBG_UPLOAD_STREAMS = 8,
       TM_DBG_UPLOAD_MERGE = 9 };
int tm_engine_debug_set_param(tm_engine *e, int param, long long value);
/* measurement hook (tools/pipeline_probe.py): two engines on one device that take turns can be CHAINED -- from now on this engine's
 * ingest stage waits for `peer`'s last column pass and its column pass for `peer`'s last row pass, so that with both engines kept
 * busy the ingest stage of one batch runs beside the row pass of the other.  peer = NULL unchains; the peer must outlive the chain. */
int tm_engine_debug_chain(tm_engine *e, tm_engine *peer);

#ifdef __cplusplus
}
#endif
#endif
