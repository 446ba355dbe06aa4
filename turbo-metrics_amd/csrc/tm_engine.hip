// tm_engine.hip -- host side of libturbometrics_hip.so: the C ABI of include/turbo_metrics_hip.h
// over the gfx950 kernels in tm_kernels.h.  No CPU compute path exists here: every entry point that
// needs the GPU returns TM_ERR_HIP / TM_ERR_UNSUPPORTED when there is none.
//
// Reference counterparts (paths relative to /root/reference/crates):
//   engine object        turbo-metrics/src/lib.rs:188-249 (TurboMetrics) + ssimulacra2-cuda/src/lib.rs:27-107
//   per-pair dataflow    turbo-metrics/src/lib.rs:268-360 (compute_one), ssimulacra2-cuda/src/lib.rs:140-229
//   post-processing      ssimulacra2-cuda/src/lib.rs:449-623
//   colour coefficients  cuda-colorspace-kernel/src/lib.rs:186-218
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/turbo_metrics_hip.h"
#include "../../include/turbo_metrics_hip_debug.h"
#include "tm_geom.h"
#include "tm_kernels.h"
#include "tm_ssim_kernels.h"
// Two builds of this file (csrc/Makefile).  `ship` (-DTM_SHIP; libturbometrics_hip.so, what the CLI and a binder link): the entry points
// of include/turbo_metrics_hip.h and nothing else are exported (ship.map), the laboratory's functions are not reachable and are dropped by
// the linker, and the straight-line reference pipeline is not compiled in.  `lab` (lab/libturbometrics_hip_lab.so, what tests/, tools/ and
// bench.py load): everything of both headers.  The kernels the two builds share are the same device code, instruction for instruction
// (tests/test_abi_symbols.py compares them).
#ifndef TM_SHIP
#include "tm_reference_kernels.h"
#endif
#include "tm_tables.inc"
#include "tm_math_tables.inc"

namespace {

thread_local char g_hip_err[256] = "";

int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
    return TM_ERR_HIP;
}
#define HIPCHK(call)                                        \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(e_, #call);   \
    } while (0)

// ---- streams and the hardware queues behind them -----------------------------------------------------------------------------------
// The runtime binds a stream to one of a handful of hardware queues (four by default) WHEN THE STREAM IS CREATED -- the least-used
// queue, the most recently opened one first --, and kernels of two streams on one queue run one after the other
// (tools/microbench/queue_map.hip, profiles/r06y6_queue_map.log).  An engine needs three streams that run beside each other: its own,
// the side stream of the fused EDGE kernel and the second upload stream.  Created first in a process they get a queue each; created
// behind somebody else's streams -- an RCCL communicator, a decoder -- they may not: with the process group of `bench.py --gpus N`
// initialised before the engine, the side stream sat on the engine's own queue, the fused kernel ran behind the column pass instead of
// beside it, and the headline went from 14.6 k to 11.7 k pairs/s (profiles/r06y9_dist_first.log).  So the library finds out where a
// stream landed and asks for another one when that place is taken:
//   lanes      per device, one ANCHOR stream per hardware queue the runtime hands out, found once by creating streams and timing pairs
//              of 100-us one-wave spin kernels on them that stamp the device's clock at their start and end (on one queue the second
//              kernel starts after the first one has ended; no host timing is involved)
//   lane_of    the anchor a new stream serialises with = its queue
//   pick       candidate streams are created (each lands on the next least-used queue) until one sits on the cheapest queue there is:
//              sharing a queue with another engine's stream costs most, then with the upload stream, then with the side stream
//              (which only engines of more than four slots ever use)
// The side stream and the second upload stream are ONE each per device and process, shared by all engines (a stream of each per
// engine would use the queues up: host-fed 4K 1 080 -> 930 pairs/s in round 4); engines that compute at the same time take turns
// on them (each launch is ordered by its own fork / join events).  Copies that follow each other on one stream leave the link idle
// between them (3-MB copies with a fence per pair: 40 GB/s on one stream, 48-52 on two; profiles/r04y_dma_depth_probe.log): page-locked
// frames of the distorted side go up on the upload stream while those of the reference side go up on the engine's own.
// TM_QUEUE_LANES=0 in the environment: streams are taken as they come (measurements).
std::atomic<int> g_debug_log{0}; // tm_set_debug_log: diagnostics on stderr (what the placement search measured, where the streams landed)
__global__ void __launch_bounds__(64) k_lane_spin(long long ticks, long long *stamp)
{
    const long long t0 = wall_clock64(); // (the device's constant 100-MHz counter: the same clock on every CU)
    long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (stamp && threadIdx.x == 0) { stamp[0] = t0; stamp[1] = t; }
}

std::mutex g_side_mutex;
struct Lanes {
    bool probed = false, off = false;
    std::vector<hipStream_t> anchor; // one stream per hardware queue seen
    std::vector<int> engines;        // engine streams per lane
    int side = -1, up = -1;          // lanes of the two shared streams (-1: none / unknown)
    long long *stamp = nullptr;      // page-locked, device-visible: [start, end] of the two kernels of a test
};
Lanes g_lanes[64];
hipStream_t g_side_stream[64] = {};
int g_side_users[64] = {};
hipStream_t g_up_stream[64] = {};
int g_up_users[64] = {};
constexpr long long LANE_TICKS = 10000; // 100 us

// do kernels on a and b run one after the other?  Read off the DEVICE's clock: the second kernel started after the first one had ended.
// (a serial pair cannot look parallel; a parallel pair on a busy device can look serial -- the second kernel found no room -- : confirmed twice)
bool lanes_serialise(Lanes &L, hipStream_t a, hipStream_t b)
{
    for (int rep = 0; rep < 2; ++rep) {
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return false; }
        L.stamp[0] = L.stamp[1] = L.stamp[2] = L.stamp[3] = 0;
        hipLaunchKernelGGL(k_lane_spin, dim3(1), dim3(64), 0, a, LANE_TICKS, L.stamp);
        hipLaunchKernelGGL(k_lane_spin, dim3(1), dim3(64), 0, b, LANE_TICKS, L.stamp + 2);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return false; }
        const volatile long long *t = L.stamp;
        if (t[1] == 0 || t[3] == 0) return false;             // (a stamp did not arrive: no verdict)
        if (!(t[2] >= t[1] || t[0] >= t[3])) return false;    // the two ran at the same time for a while
    }
    return true;
}

// (g_side_mutex held)  the queues this process gets streams on, one anchor each
void lanes_probe(int device)
{
    Lanes &L = g_lanes[device];
    if (L.probed) return;
    L.probed = true;
    const char *env = getenv("TM_QUEUE_LANES");
    if (env && atoi(env) == 0) { L.off = true; return; }
    hipStream_t warm = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    if (hipHostMalloc((void **)&L.stamp, 4 * sizeof(long long), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); L.stamp = nullptr; L.off = true; return; }
    if (hipStreamCreateWithFlags(&warm, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); L.off = true; return; }
    hipLaunchKernelGGL(k_lane_spin, dim3(1), dim3(64), 0, warm, 100, (long long *)nullptr); // the first launch loads the code object
    if (hipStreamSynchronize(warm) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamDestroy(warm); L.off = true; return; }
    std::vector<hipStream_t> extra;
    const auto t_warm = std::chrono::steady_clock::now();
    L.anchor.push_back(warm);
    for (int i = 0, repeats = 0; i < 12 && repeats < 2 && L.anchor.size() < 8; ++i) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        bool seen = false;
        for (hipStream_t a : L.anchor) if (lanes_serialise(L, s, a)) { seen = true; break; }
        if (seen) { extra.push_back(s); ++repeats; } // (kept alive until the end: the next stream then goes to another queue)
        else { L.anchor.push_back(s); repeats = 0; }
    }
    for (hipStream_t s : extra) (void)hipStreamDestroy(s);
    L.engines.assign(L.anchor.size(), 0);
    if (g_debug_log.load()) fprintf(stderr, "[tm] device %d: %zu hardware queues found in %.1f ms (+ %.1f ms for the first stream and launch of the process)\n", device, L.anchor.size(),
                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_warm).count(), std::chrono::duration<double, std::milli>(t_warm - t_begin).count());
    if (L.anchor.size() < 2) L.off = true; // one queue: nothing to choose
}
int lane_of(Lanes &L, hipStream_t s)
{
    for (size_t i = 0; i < L.anchor.size(); ++i) if (lanes_serialise(L, s, L.anchor[i])) return (int)i;
    return -1; // a queue no anchor sits on (the runtime opened another one): nothing known shares it
}
enum LaneRole { LANE_ENGINE_SMALL, LANE_ENGINE, LANE_SIDE, LANE_UP };
int lane_cost(const Lanes &L, int lane, LaneRole role)
{
    if (lane < 0) return 0;
    int c = 8 * L.engines[(size_t)lane];
    if (role != LANE_UP && lane == L.up) c += 4;
    if (role != LANE_SIDE && lane == L.side) c += role == LANE_ENGINE_SMALL ? 1 : 4;
    return c;
}
// (g_side_mutex held)  a new stream for `role`, on the cheapest queue of up to six tries
hipStream_t lanes_pick(int device, LaneRole role, int *lane_out)
{
    *lane_out = -1;
    if (device < 0 || device >= 64) return nullptr;
    lanes_probe(device);
    Lanes &L = g_lanes[device];
    hipStream_t best = nullptr;
    int best_cost = 1 << 30, best_lane = -1, reachable = 1 << 30; // reachable: the cheapest lane there is -- no candidate can do better
    for (size_t l = 0; l < L.anchor.size(); ++l) reachable = std::min(reachable, lane_cost(L, (int)l, role));
    std::vector<hipStream_t> rejected;
    for (int t = 0; t < (L.off ? 1 : (int)L.anchor.size() + 2); ++t) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        const int lane = L.off ? -1 : lane_of(L, s);
        const int cost = L.off ? 0 : lane_cost(L, lane, role);
        if (cost < best_cost) { if (best) rejected.push_back(best); best = s; best_cost = cost; best_lane = lane; }
        else rejected.push_back(s);
        if (best_cost <= reachable) break;
    }
    if (g_debug_log.load()) fprintf(stderr, "[tm] device %d: stream for %s on hardware queue %d of %zu (cost %d, %zu candidate(s) given back)\n", device,
                                    role == LANE_SIDE ? "the fused kernel" : role == LANE_UP ? "uploads" : role == LANE_ENGINE ? "an engine" : "a small engine", best_lane, L.anchor.size(), best ? best_cost : -1, rejected.size());
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    *lane_out = best_lane;
    return best;
}
hipStream_t engine_stream_acquire(int device, bool small, int *lane_out)
{
    std::lock_guard<std::mutex> lock(g_side_mutex);
    hipStream_t s = lanes_pick(device, small ? LANE_ENGINE_SMALL : LANE_ENGINE, lane_out);
    if (s && *lane_out >= 0) ++g_lanes[device].engines[(size_t)*lane_out];
    return s;
}
void engine_stream_release(int device, hipStream_t s, int lane)
{
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (device >= 0 && device < 64 && lane >= 0 && (size_t)lane < g_lanes[device].engines.size() && g_lanes[device].engines[(size_t)lane] > 0) --g_lanes[device].engines[(size_t)lane];
    if (s) (void)hipStreamDestroy(s);
}
hipStream_t side_stream_acquire(int device)
{
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (device < 0 || device >= 64) return nullptr;
    if (!g_side_stream[device] && !(g_side_stream[device] = lanes_pick(device, LANE_SIDE, &g_lanes[device].side))) return nullptr;
    ++g_side_users[device];
    return g_side_stream[device];
}
void side_stream_release(int device)
{
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (device < 0 || device >= 64 || g_side_users[device] <= 0) return;
    if (--g_side_users[device] == 0 && g_side_stream[device]) { (void)hipStreamSynchronize(g_side_stream[device]); (void)hipStreamDestroy(g_side_stream[device]); g_side_stream[device] = nullptr; g_lanes[device].side = -1; }
}
hipStream_t up_stream_acquire(int device)
{
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (device < 0 || device >= 64) return nullptr;
    if (!g_up_stream[device] && !(g_up_stream[device] = lanes_pick(device, LANE_UP, &g_lanes[device].up))) return nullptr;
    ++g_up_users[device];
    return g_up_stream[device];
}
void up_stream_release(int device)
{
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (device < 0 || device >= 64 || g_up_users[device] <= 0) return;
    if (--g_up_users[device] == 0 && g_up_stream[device]) { (void)hipStreamSynchronize(g_up_stream[device]); (void)hipStreamDestroy(g_up_stream[device]); g_up_stream[device] = nullptr; g_lanes[device].up = -1; }
}

// engines alive in this process (tm_engine_debug_chain keeps a raw pointer to a peer: destroying the peer must unhook it)
std::vector<tm_engine *> g_engines;

const uint32_t k_lut_bits[256] = {TM_SRGB_LUT_BITS};
const double k_weights[108] = {TM_SSIMU2_WEIGHTS};
// the math table buffer of tm_device_math.h (TM_TAB_DOUBLES): pow_pos tables, then the binary64 transfer-function cubics
struct TmMathTab {
    double pow[TM_TAB_POW_DOUBLES];
    double eotf64[TM_EOTF64_SEGS * TM_EOTF64_STRIDE];
};
static_assert(sizeof(TmMathTab) == TM_TAB_DOUBLES * sizeof(double) && offsetof(TmMathTab, eotf64) == TM_TAB_EOTF64 * sizeof(double), "math table layout");
const TmMathTab k_powtab = {{TM_POW_RCP, TM_POW_NLOG, TM_POW_EXP2}, {TM_EOTF64_C}};

// ---- colour coefficients: same f32 operation order as the reference's const evaluation -----------
struct V3 { float x, y, z; };
V3 xy_to_xyz(float x, float y) { return {x / y, 1.0f, (1.0f - x - y) / y}; } // const_algebra.rs:30-32
float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

void kr_kb(int matrix, float *kr, float *kb) // lib.rs:203-218 + constants.rs:3-18
{
    static const float prim[3][8] = {
        {0.640f, 0.330f, 0.300f, 0.600f, 0.150f, 0.060f, 0.3127f, 0.3290f}, // BT709
        {0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f}, // BT601_525
        {0.640f, 0.330f, 0.290f, 0.600f, 0.150f, 0.060f, 0.3127f, 0.3290f}, // BT601_625
    };
    const float *p = prim[matrix];
    const V3 r = xy_to_xyz(p[0], p[1]), g = xy_to_xyz(p[2], p[3]), b = xy_to_xyz(p[4], p[5]),
             w = xy_to_xyz(p[6], p[7]);
    const V3 xr = {r.x, g.x, b.x}, yr = {r.y, g.y, b.y}, zr = {r.z, g.z, b.z};
    const float mul = 1.0f / dot3(xr, cross3(yr, zr));
    *kr = dot3(w, cross3(g, b)) * mul;
    *kb = dot3(w, cross3(r, g)) * mul;
}

void yuv_coefficients(int matrix, int bits, float out[5]) // lib.rs:186-200, Limited range :103-132
{
    float kr, kb;
    kr_kb(matrix, &kr, &kb);
    const float luma_range = (float)((235u - 16u) << (bits - 8));
    const float chroma_range = (float)((240u - 16u) << (bits - 8));
    const float kg = 1.0f - kr - kb;
    out[0] = 1.0f / luma_range;
    out[1] = 2.0f * (1.0f - kr) * 1.0f / chroma_range;
    out[2] = 2.0f * (1.0f - kb) * 1.0f / chroma_range;
    out[3] = -2.0f * (1.0f - kb) * kb / kg * 1.0f / chroma_range;
    out[4] = -2.0f * (1.0f - kr) * kr / kg * 1.0f / chroma_range;
}

} // namespace

struct tm_engine {
    int device = 0;
    uint32_t w = 0, h = 0, mask = 0, cap = 0;
    TmGeom g{};
    TmJobs jobs{};                 // job table of the two blur passes (EDGE jobs inside their scale)
    TmJobs jobs_f{};               // the same jobs with the EDGE jobs last: launches whose EDGE jobs run in k_blur_edge_fused
    unsigned long long *HS = nullptr; // fused EDGE kernel: state hand-off words [plane][2][ef_tiles][6][64]
    double *EROWS = nullptr;          // ... per-row sums [plane][ef_bands][64][2]
    unsigned *d_epoch = nullptr;      // ... launch epoch of the hand-off tags (advanced by k_finish_edge)
    int *d_status = nullptr, *h_status = nullptr; // ... a hand-off wait that timed out
    int ef_tiles = 0, ef_bands = 0, ef_ne = 0;
    size_t hs_elems = 0, erows_elems = 0; // what HS / EROWS hold (mem_bytes bookkeeping; the geometry above is what the kernels index with)
    unsigned ef_epoch_host = 1;      // host mirror of d_epoch[0] (one step per fused launch): HS is cleared when the 24-bit tag epoch wraps
    bool ef_epoch_unknown = false;   // a fused launch failed part-way or bailed out: the mirror is re-read from the device before the next one
    // The tuning values below are fixed in a release build's behaviour: no environment variable reaches them.  tools/ and tests move
    // them through tm_engine_debug_set_param (TM_DBG_*), per engine.
    long long fused_edge_from = 400; // bands of 32 rows of EDGE planes per launch (slots x jobs x ceil(h / 32)) from which those jobs take the fused kernel: 6 pairs of
                                     // 1080p, 3 of 4K, 9 of 720p (below, the launch is bound by the latency of one wave walking its band and of the chain of bands; round 3: 340 --
                                     // with round 4's eight-wave row pass the two passes win at 5 pairs of 1080p, 9.4 k vs 8.4 k pairs/s: profiles/r04v_fused_threshold.log)
    int ef_beside = 1;  // the fused kernel runs on stream2 beside the two blur passes: 1 = enqueued before the column pass, 2 = after it, 0 = behind the row pass on the engine's stream
    int ef_waves = 4;   // waves per workgroup of the fused kernel
    int ef_persist_wgs = 0; // workgroups of the fused kernel when it runs beside the passes: 0 = 7/8 per CU, > 0 = that many, < 0 = one per ticket
    int ef_pass_prio = 1;   // the two passes raise their waves' issue priority while the fused kernel runs beside them
    int n_cus = 256;
    int dbg_no_linear_upload = 1; // TM_DBG_LINEAR_UPLOAD 0 (default): tight planar host pictures as 2-D copies into padded rows, like any other; 1: one linear copy
    int ef_fault = 0;   // fault injection (TM_DBG_EF_FAULT): 1 = do not wait for the band above, 2 = do not publish the state (the hand-off then times out: TM_ERR_HIP from tm_engine_sync)
    bool full_sums = false;
    int channel_mode = TM_CHANNELS_POOLED;
    bool use_graph = false;         // THIS launch replays its sequence from a captured hipGraph (decided per launch in tm_engine_compute_async)
    int graph_mode = -1;            // tm_engine_set_graph: 1 always, 0 never; -1 (default) = by itself for small launches that repeat (below)
    long long auto_key = -1;        // ... the launch shape seen last and how often in a row
    int auto_seen = 0;
    int lane = -1;                  // the hardware queue its stream sits on (g_lanes)
    hipGraphExec_t gexec = nullptr;
    long long gkey = -1;
    TmSsimGeom sg{};               // SSIM / MS-SSIM (only when the mask asks for them)
    unsigned char *QU8 = nullptr;  // [slot][side][3] planar u8-quantised linear RGB
    unsigned short *SPYR = nullptr; // [slot][side][3] box-sum pyramid, scales 1..4
    double *SPART = nullptr, *SSUMS = nullptr, *h_ssums = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // the fused EDGE kernel beside the two blur passes (fork after the ingest stage, join before the finisher)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_col_done = nullptr, ev_row_done = nullptr; // recorded behind the column pass / the row pass of every launch (an engine chained to this one waits for them)
    tm_engine *chain_peer = nullptr; // tm_engine_debug_chain: this engine's ingest stage waits for the peer's column pass, its column pass for the peer's row pass
    float *LIN = nullptr, *XYB = nullptr, *XYBT = nullptr, *V = nullptr; // LIN, XYBT: reference pipeline only (TM_VARIANT_REFERENCE)
    float *V_alloc = nullptr; // the allocation behind V (V = V_alloc + an offset inside TM_V_SLACK, see tm_engine_debug_set_v_offset)
    float *LIN2 = nullptr; // level-2 linear RGB [slot][side][3 planes]: hand-off k_ingest_wave -> k_ingest_upper_rd
    double *PART = nullptr, *SUMS = nullptr;
    unsigned long long *SSE = nullptr;
    TmFrameDesc *d_desc = nullptr, *h_desc = nullptr;
    double *h_sums = nullptr;
    unsigned long long *h_sse = nullptr;
    float *d_lut = nullptr, *d_coef = nullptr;
    double *d_powtab = nullptr;
    std::vector<hipEvent_t> up_ev;    // upload fences (tm_engine_upload_fence): a small ring of events on the engine's stream
    std::vector<hipEvent_t> up_ev2;   // ... and on the second upload stream, for the fences that had copies on it
    std::vector<uint64_t> up_tok2;    // per fence: token of the most recent fence (itself included) that recorded on the second upload stream, or UINT64_MAX
    uint64_t up_last2 = UINT64_MAX;   // token of the most recent fence that recorded an event on the second upload stream
    hipStream_t up_stream = nullptr;  // the device's second upload stream (page-locked frames of the distorted side)
    hipEvent_t ev_up_join = nullptr, ev_stage_free = nullptr;
    bool up_pending = false;          // copies on up_stream that the engine's stream has not been made to wait for yet
    bool up_since_fence = false;      // copies on up_stream since the last fence
    bool stage_busy = false;          // a launch may still be reading the staging surfaces: up_stream waits for ev_stage_free first
    int upload_streams = 2;           // TM_DBG_UPLOAD_STREAMS
    uint64_t up_next = 0;             // tokens handed out so far (token t lives in up_ev[t % size] until token t + size is taken)
    std::vector<void *> staging;      // [slot*2+side], lazily allocated: a slot of the side's arena (below) or an allocation of its own
    std::vector<size_t> staging_size;
    std::vector<char> staging_own;    // [slot*2+side]: 1 = its own hipMalloc (a frame that does not fit the arena's slots)
    // Round 6: the staging surfaces of a side are ONE allocation, slot after slot at the size of the first frame handed over, so that page-locked
    // frames that lie back to back in the caller's memory (a decoder's surface pool, the CLI's frame ring) land back to back here too -- and two
    // of them go up as ONE DMA (pend: a copy is held back until the next one of its stream is known).  Beside the kernels 3-MB copies on the
    // two upload streams reach 48 GB/s, 6-MB ones 52-56, 25-MB ones 56-57 (profiles/r06j_dma_size_probe.log).
    char *stage_arena[2] = {nullptr, nullptr};
    size_t stage_stride[2] = {0, 0};
    struct PendingCopy { char *dst; const char *src; size_t dpitch, spitch, row_bytes, rows; bool linear; };
    PendingCopy pend[2] = {};          // [0] engine's stream, [1] second upload stream
    bool has_pend[2] = {false, false};
    size_t merge_limit = (size_t)14 << 20; // bytes up to which copies are merged (1080p: up to four frames; 4K frames go up one by one)
    size_t mem_bytes = 0;
    bool profiling = false, ev_pending = false;
    hipEvent_t ev[7] = {}; // 0..4: start | ingest | column pass | row pass | finisher + SSIM stage, on the engine's stream; 5, 6: around the fused EDGE kernel, on the stream it runs on
    bool edge_timed = false; // the last profiled launch ran the fused kernel (events 5, 6 were recorded)
    double stage_ms[TM_STAGE_COUNT] = {0, 0, 0, 0, 0};
    uint64_t n_prof = 0;
    uint32_t last_n = 0;
    bool in_flight = false, have_results = false;
    int variant = TM_VARIANT_DEFAULT;
    long long split_rows_below = 1024; // row blocks per launch up to which the eight-wave row pass runs (2 600 beside the fused kernel, whose passes hold the FULL jobs only: see launch_batch)
    bool split_rows_env = false; // TM_DBG_SPLIT_ROWS_BELOW was set: used as it is
    long long solo_col_below = 800; // role-waves of the column pass per launch (5 per column block) up to which each runs as its own workgroup: one 1080p pair (745; two pairs lose)
    int ingest_rows = 0; // quad rows per wave of k_ingest_rows; 0 = chosen per launch (tm_engine_debug_set_ingest_rows)
};

namespace {

template <typename T> int dev_alloc(tm_engine *e, T **p, size_t count, bool zero)
{
    const size_t bytes = count * sizeof(T);
    hipError_t r = hipMalloc((void **)p, bytes ? bytes : 1);
    if (r == hipErrorOutOfMemory) { snprintf(g_hip_err, sizeof g_hip_err, "hipMalloc(%zu): out of memory", bytes); return TM_ERR_OOM; }
    if (r != hipSuccess) return hip_fail(r, "hipMalloc");
    e->mem_bytes += bytes;
    if (zero) {
        // hipMemset on device memory returns before the fill has run (it is a kernel on the null stream), and the engine's streams
        // are non-blocking ones that do not wait for the null stream: a launch enqueued right after an allocation made later than
        // tm_engine_create (TM_VARIANT_REFERENCE's arenas, grown hand-off buffers) had its first results zeroed under it
        // (tests/soak/variant_sweep_soak.py found it: the first launch after the switch, large frames only).  Wait for the fill.
        HIPCHK(hipMemset(*p, 0, bytes ? bytes : 1));
        HIPCHK(hipStreamSynchronize(nullptr));
    }
    return TM_OK;
}

// an engine lives on the device that was current when it was created; calls may come from a thread whose current device
// is another one (one process driving several GPUs)
#define TM_BIND(e) do { if ((e) && hipSetDevice((e)->device) != hipSuccess) return hip_fail(hipGetLastError(), "hipSetDevice"); } while (0)

int check_slot_side(const tm_engine *e, uint32_t slot, int side)
{
    if (!e || slot >= e->cap || (side != TM_SIDE_REF && side != TM_SIDE_DIS)) return TM_ERR_INVALID_ARG;
    return TM_OK;
}

int submit_copy(tm_engine *e, int si, const tm_engine::PendingCopy &c)
{
    hipStream_t st = si ? e->up_stream : e->stream;
    if (c.linear) HIPCHK(hipMemcpyAsync(c.dst, c.src, c.row_bytes, hipMemcpyHostToDevice, st));
    else HIPCHK(hipMemcpy2DAsync(c.dst, c.dpitch, c.src, c.spitch, c.row_bytes, c.rows, hipMemcpyHostToDevice, st));
    return TM_OK;
}

// the copy that is held back on stream si (if any) goes out
int flush_pending(tm_engine *e, int si)
{
    if (!e->has_pend[si]) return TM_OK;
    e->has_pend[si] = false;
    return submit_copy(e, si, e->pend[si]);
}
int flush_pending(tm_engine *e)
{
    int rc = flush_pending(e, 0);
    const int rc2 = flush_pending(e, 1);
    return rc ? rc : rc2;
}

// A page-locked copy for stream si: merged with the one held back when the two are one run of bytes (or of rows) on both ends -- then the pair
// goes out as ONE DMA --, otherwise the held one goes out and this one is held.  Never more than one copy is held per stream, and every
// fence, launch and sync flushes: the caller's "bytes stay untouched until the fence is done / tm_engine_sync returns" does not change.
int queue_copy(tm_engine *e, int si, const tm_engine::PendingCopy &c)
{
    const size_t bytes = c.linear ? c.row_bytes : c.row_bytes * c.rows;
    if (e->has_pend[si]) {
        tm_engine::PendingCopy &p = e->pend[si];
        const size_t pbytes = p.linear ? p.row_bytes : p.row_bytes * p.rows;
        const bool run = p.linear == c.linear && pbytes + bytes <= e->merge_limit &&
                         (c.linear ? c.src == p.src + p.row_bytes && c.dst == p.dst + p.row_bytes
                                   : c.row_bytes == p.row_bytes && c.spitch == p.spitch && c.dpitch == p.dpitch && c.src == p.src + p.spitch * p.rows && c.dst == p.dst + p.dpitch * p.rows);
        if (run) {
            if (c.linear) p.row_bytes += c.row_bytes; else p.rows += c.rows;
            // held on while one more frame of this size would still fit (1080p: up to four frames = 13 MB per DMA; beside the kernels 3 / 6 / 12 / 25 MB
            // copies reach 48 / 52-56 / 54-56 / 56-57 GB/s)
            return pbytes + 2 * bytes <= e->merge_limit ? TM_OK : flush_pending(e, si);
        }
        const int rc = flush_pending(e, si);
        if (rc) return rc;
    }
    if (2 * bytes > e->merge_limit) return submit_copy(e, si, c); // too large to share a DMA with a second one of its size
    e->pend[si] = c;
    e->has_pend[si] = true;
    return TM_OK;
}

// copy `rows` rows of `row_bytes` from host memory (pitch `src_pitch`) into the staging surface; a page-locked source may be held back for a
// moment (queue_copy), anything else goes out now, behind whatever was held on its stream
int stage_rows(tm_engine *e, hipStream_t st, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes, size_t rows, bool pinned = false)
{
    if (rows == 0 || row_bytes == 0) return TM_OK;
    const int si = st == e->up_stream && e->up_stream ? 1 : 0;
    const tm_engine::PendingCopy c{(char *)dst, (const char *)src, dst_pitch, src_pitch, row_bytes, rows, false};
    if (pinned) return queue_copy(e, si, c);
    const int rc = flush_pending(e, si);
    return rc ? rc : submit_copy(e, si, c);
}

// the stream a host frame goes up on: the engine's own, or -- page-locked frames of the distorted side -- the device's second upload
// stream, which first waits for the last launch that read the staging surfaces; tm_engine_compute_async makes the engine's stream wait
// for these copies, fences and tm_engine_sync cover both streams
int upload_stream(tm_engine *e, int side, int mem, hipStream_t *out)
{
    *out = e->stream;
    if (mem != TM_MEM_HOST_PINNED || side != TM_SIDE_DIS || !e->up_stream || e->upload_streams < 2) {
        // a distorted-side copy on the engine's own stream after page-locked ones on the second stream: same staging surface, two
        // streams -- the engine's stream waits for the second one first (what tm_engine_compute_async would do later anyway)
        if (side == TM_SIDE_DIS && mem != TM_MEM_DEVICE && e->up_pending) {
            { const int rc = flush_pending(e, 1); if (rc) return rc; } // (a copy still held back belongs in front of the join)
            HIPCHK(hipEventRecord(e->ev_up_join, e->up_stream));
            HIPCHK(hipStreamWaitEvent(e->stream, e->ev_up_join, 0));
            e->up_pending = false;
        }
        return TM_OK;
    }
    if (e->stage_busy) {
        HIPCHK(hipStreamWaitEvent(e->up_stream, e->ev_stage_free, 0));
        e->stage_busy = false;
    }
    e->up_pending = e->up_since_fence = true;
    *out = e->up_stream;
    return TM_OK;
}

int ensure_staging(tm_engine *e, size_t idx, size_t bytes)
{
    if (e->staging_size[idx] >= bytes) return TM_OK;
    const int side = (int)(idx & 1);
    const size_t stride = (bytes + 255) / 256 * 256;
    // the side's arena: made for the first frame handed over, slots of exactly that frame's size (capped: a rarely used 4K engine of 128 slots
    // does not get 3 GB of staging it may never fill -- its frames are large copies anyway)
    if (!e->stage_arena[side] && (size_t)e->cap * stride <= ((size_t)2 << 30)) {
        hipError_t r = hipMalloc((void **)&e->stage_arena[side], (size_t)e->cap * stride);
        if (r == hipSuccess) { e->stage_stride[side] = stride; e->mem_bytes += (size_t)e->cap * stride; }
        else { (void)hipGetLastError(); e->stage_arena[side] = nullptr; } // (the slot gets an allocation of its own below, or fails there)
    }
    if (e->stage_arena[side] && e->stage_stride[side] >= bytes && !e->staging_own[idx]) {
        e->staging[idx] = e->stage_arena[side] + (idx >> 1) * e->stage_stride[side];
        e->staging_size[idx] = e->stage_stride[side];
        return TM_OK;
    }
    // a frame that does not fit the arena's slots (another kind, another pitch): an allocation of its own, grown when needed
    int rc = flush_pending(e);
    if (rc) return rc;
    if (e->staging[idx]) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->up_pending) HIPCHK(hipStreamSynchronize(e->up_stream));
        if (e->staging_own[idx]) { HIPCHK(hipFree(e->staging[idx])); e->mem_bytes -= e->staging_size[idx]; }
        e->staging[idx] = nullptr; e->staging_size[idx] = 0; e->staging_own[idx] = 0;
    }
    hipError_t r = hipMalloc(&e->staging[idx], bytes);
    if (r == hipErrorOutOfMemory) return TM_ERR_OOM;
    if (r != hipSuccess) return hip_fail(r, "hipMalloc(staging)");
    e->staging_size[idx] = bytes; e->staging_own[idx] = 1; e->mem_bytes += bytes;
    return TM_OK;
}

// coded_rows: 0 = two planes (p0, p1); > 0 = ONE surface declared by the caller: `coded_rows` luma rows at p0, then the CbCr rows
// (tm_engine_set_surface_*: the reference's decoded-surface contract, cudarse-video/src/dec.rs:299-393)
int set_frame_common(tm_engine *e, uint32_t slot, int side, int kind, const void *p0, const void *p1, size_t pitch,
                     int matrix, int mem, size_t coded_rows = 0)
{
    int rc = check_slot_side(e, slot, side);
    if (rc) return rc;
    TM_BIND(e);
    if (!p0 || (mem != TM_MEM_HOST && mem != TM_MEM_DEVICE && mem != TM_MEM_HOST_PINNED)) return TM_ERR_INVALID_ARG;
    const bool yuv = kind == TM_KIND_NV12 || kind == TM_KIND_P016;
    if (yuv && !p1) return TM_ERR_INVALID_ARG;
    const size_t bps = kind == TM_KIND_NV12 || kind == TM_KIND_RGB8 ? 1 : (kind == TM_KIND_P016 || kind == TM_KIND_RGB16 ? 2 : 4);
    const size_t row_bytes = yuv ? (size_t)e->w * bps : (size_t)e->w * 3 * bps;
    if (pitch < row_bytes) return TM_ERR_INVALID_ARG;
    // the ingest kernel addresses a surface with 32-bit lane offsets (row * pitch through a 24-bit multiply)
    if (pitch >= ((size_t)1 << 24) || pitch * ((size_t)e->h + (e->h + 1) / 2) >= ((size_t)1 << 32)) return TM_ERR_INVALID_ARG;
    TmFrameDesc &d = e->h_desc[slot * 2 + side];
    if (e->in_flight) { // descriptors are read by an async copy; do not race with it
        rc = tm_engine_sync(e);
        if (rc) return rc;
    }
    if (mem == TM_MEM_DEVICE) {
        d.p0 = p0; d.p1 = p1; d.pitch = pitch;
    } else {
        const size_t idx = slot * 2 + side;
        const size_t spitch = (row_bytes + 255) / 256 * 256;
        const size_t chroma_rows = (e->h + 1) / 2;
        const size_t need = yuv ? spitch * ((coded_rows > e->h ? coded_rows : e->h) + chroma_rows) : spitch * e->h; // room for a declared surface's padding rows (below)
        rc = ensure_staging(e, idx, need);
        if (rc) return rc;
        hipStream_t us;
        if ((rc = upload_stream(e, side, mem, &us))) return rc;
        const bool pinned = mem == TM_MEM_HOST_PINNED;
        char *s = (char *)e->staging[idx];
        const size_t uv_bytes = yuv ? (size_t)((e->w + 1) / 2) * 2 * bps : 0, uv_row = uv_bytes <= pitch ? uv_bytes : pitch;
        // ONE 2-D copy of all rows instead of two (the per-copy cost is what limits small frames: docs/LABBOOK.md section 5, host-fed) in
        // exactly two cases, neither of which reads a byte the caller has not declared:
        //   * the caller declared a surface (tm_engine_set_surface_*: coded_rows luma rows, the padding rows included, then the CbCr rows);
        //   * the CbCr rows start exactly where the h luma rows end (gap == pitch * h): every row of the copy belongs to one of the two planes.
        // Round 3 inferred "one allocation" from any distance of h .. h + 64 rows and then read the rows in between.
        const size_t gap = yuv && (const char *)p1 >= (const char *)p0 ? (size_t)((const char *)p1 - (const char *)p0) : 0;
        const size_t luma_rows = coded_rows ? coded_rows : (yuv && gap == pitch * (size_t)e->h ? (size_t)e->h : 0);
        if (luma_rows >= e->h) {
            rc = stage_rows(e, us, s, spitch, p0, pitch, row_bytes > uv_row ? row_bytes : uv_row, luma_rows + chroma_rows, pinned);
            if (rc) return rc;
            d.p1 = s + spitch * luma_rows;
        } else {
            rc = stage_rows(e, us, s, spitch, p0, pitch, row_bytes, e->h, pinned);
            if (rc) return rc;
            if (yuv) {
                rc = stage_rows(e, us, s + spitch * e->h, spitch, p1, pitch, uv_row, chroma_rows, pinned);
                if (rc) return rc;
                d.p1 = s + spitch * e->h;
            } else d.p1 = nullptr;
        }
        // pageable source: make sure the bytes have left the caller's buffer before returning (a pinned source is
        // the caller's to keep alive until tm_engine_sync, so its DMA stays asynchronous)
        if (mem == TM_MEM_HOST) HIPCHK(hipStreamSynchronize(e->stream));
        d.p0 = s; d.pitch = spitch;
    }
    d.kind = kind; d.matrix = matrix; d.p2 = nullptr; d.pitch2 = 0; d.shift = 0;
    return TM_OK;
}

// planar 4:2:0 (TM_KIND_I420_8 / I420_16): three planes; the staged copy keeps them planar (Y rows, then Cb rows, then Cr rows)
// packed10: TM_KIND_I420_P10 -- three 10-bit samples per 32-bit word (tm_geom.h), rows of whole 512-byte blocks
int set_frame_planar(tm_engine *e, uint32_t slot, int side, const void *y, const void *u, const void *v, size_t pitch_y,
                     size_t pitch_uv, int bits, int matrix, int mem, bool packed10 = false)
{
    int rc = check_slot_side(e, slot, side);
    if (rc) return rc;
    TM_BIND(e);
    if (!y || !u || !v || (mem != TM_MEM_HOST && mem != TM_MEM_DEVICE && mem != TM_MEM_HOST_PINNED)) return TM_ERR_INVALID_ARG;
    if (bits < 8 || bits > 16) return TM_ERR_INVALID_ARG;
    const size_t bps = bits == 8 ? 1 : 2;
    const size_t cw = (e->w + 1) / 2, ch = (e->h + 1) / 2;
    const size_t row_y = packed10 ? (size_t)tm_p10_row_words(e->w) * 4 : (size_t)e->w * bps, row_c = packed10 ? (size_t)tm_p10_row_words(cw) * 4 : cw * bps;
    if (packed10 && (((uintptr_t)y | (uintptr_t)u | (uintptr_t)v | pitch_y | pitch_uv) & 7)) return TM_ERR_INVALID_ARG; // the kernels read word pairs
    const int kind = packed10 ? TM_KIND_I420_P10 : (bits == 8 ? TM_KIND_I420_8 : TM_KIND_I420_16);
    if (pitch_y < row_y || pitch_uv < row_c) return TM_ERR_INVALID_ARG;
    // the kernels address a plane with 32-bit lane offsets (row * pitch through a 24-bit multiply), chroma planes included
    if (pitch_y >= ((size_t)1 << 24) || pitch_y * (size_t)e->h >= ((size_t)1 << 32) || pitch_uv >= ((size_t)1 << 24) || pitch_uv * ch >= ((size_t)1 << 32)) return TM_ERR_INVALID_ARG;
    TmFrameDesc &d = e->h_desc[slot * 2 + side];
    if (e->in_flight) { // descriptors are read by an async copy; do not race with it
        rc = tm_engine_sync(e);
        if (rc) return rc;
    }
    if (mem == TM_MEM_DEVICE) {
        d.p0 = y; d.p1 = u; d.p2 = v; d.pitch = pitch_y; d.pitch2 = pitch_uv;
    } else {
        const size_t idx = slot * 2 + side;
        const size_t sp_y = (row_y + 255) / 256 * 256, sp_c = (row_c + 255) / 256 * 256;
        // (a tight picture that goes up as ONE linear copy gets a slot of exactly its size: pictures back to back in the caller's ring then are back
        // to back here, and two of them share a DMA)
        const bool tight = pitch_y == row_y && pitch_uv == row_c && (const char *)u == (const char *)y + row_y * e->h && (const char *)v == (const char *)u + row_c * ch &&
                           row_c % 4 == 0 && !e->dbg_no_linear_upload;
        rc = ensure_staging(e, idx, tight ? row_y * e->h + 2 * row_c * ch : sp_y * e->h + 2 * sp_c * ch);
        if (rc) return rc;
        hipStream_t us;
        if ((rc = upload_stream(e, side, mem, &us))) return rc;
        const bool pinned = mem == TM_MEM_HOST_PINNED;
        char *s = (char *)e->staging[idx];
        // a TIGHT picture (rows without padding, Cb behind Y, Cr behind Cb: a picture of a Y4M / raw planar file as it lies in the file)
        // can go up as ONE linear copy and be read with its own pitches (TM_DBG_LINEAR_UPLOAD = 1).  Which way is faster depends on how the
        // caller submits: pictures handed over one by one as they arrive, with a fence per pair (the CLI) -- linear 7.1-7.7 k pairs/s of 1080p
        // against 5.3-6.1 k with the 2-D copies; a whole batch queued at once (bench.py's host_fed loop) -- 2-D 6.6 k against 5.3 k linear
        // (tools/cli_ab.sh, tools/host_fed_ab.py).  The CLI's host layer switches it on; the default is the 2-D path.
        if (tight) {
            {
                const int si = us == e->up_stream && e->up_stream ? 1 : 0;
                const tm_engine::PendingCopy c{s, (const char *)y, 0, 0, row_y * e->h + 2 * row_c * ch, 1, true};
                if (pinned) rc = queue_copy(e, si, c);
                else if (!(rc = flush_pending(e, si))) rc = submit_copy(e, si, c);
                if (rc) return rc;
            }
            if (mem == TM_MEM_HOST) HIPCHK(hipStreamSynchronize(e->stream));
            d.p0 = s; d.p1 = s + row_y * e->h; d.p2 = s + row_y * e->h + row_c * ch; d.pitch = row_y; d.pitch2 = row_c;
            d.kind = kind;
            d.matrix = matrix;
            d.shift = bits == 8 ? 0 : 16 - bits;
            return TM_OK;
        }
        char *su = s + sp_y * e->h, *sv = su + sp_c * ch;
        if ((rc = stage_rows(e, us, s, sp_y, y, pitch_y, row_y, e->h, pinned))) return rc;
        if ((const char *)v == (const char *)u + pitch_uv * ch) { // Cr follows Cb (a picture of a planar file): one copy for both
            if ((rc = stage_rows(e, us, su, sp_c, u, pitch_uv, row_c, 2 * ch, pinned))) return rc;
        } else {
            if ((rc = stage_rows(e, us, su, sp_c, u, pitch_uv, row_c, ch, pinned))) return rc;
            if ((rc = stage_rows(e, us, sv, sp_c, v, pitch_uv, row_c, ch, pinned))) return rc;
        }
        if (mem == TM_MEM_HOST) HIPCHK(hipStreamSynchronize(e->stream));
        d.p0 = s; d.p1 = su; d.p2 = sv; d.pitch = sp_y; d.pitch2 = sp_c;
    }
    d.kind = kind;
    d.matrix = matrix;
    d.shift = bits == 8 ? 0 : 16 - bits;
    return TM_OK;
}

int check_yuv_args(int matrix, int transfer, int full_range)
{
    if (matrix < 0 || matrix > 2) return TM_ERR_INVALID_ARG;
    // cuda-colorspace/src/lib.rs:45-52: full range and non-BT709 transfer are todo!() in the reference
    if (transfer != TM_TRANSFER_BT709 || full_range) return TM_ERR_UNSUPPORTED;
    return TM_OK;
}

unsigned long long slot_sse_channel(const tm_engine *e, uint32_t slot, int c)
{
    unsigned long long tot = 0;
    for (int i = 0; i < TM_SSE_BINS; ++i) tot += e->h_sse[((size_t)slot * TM_SSE_BINS + i) * 3 + c];
    return tot;
}

unsigned long long slot_sse(const tm_engine *e, uint32_t slot)
{
    return slot_sse_channel(e, slot, 0) + slot_sse_channel(e, slot, 1) + slot_sse_channel(e, slot, 2);
}

dim3 grid2(int w, int h, int z) { return dim3((unsigned)((w + 63) / 64), (unsigned)h, (unsigned)z); }

// both orderings of the job table, and the buffers of the fused EDGE kernel for the EDGE jobs the table has
int make_job_tables(tm_engine *e)
{
    tm_make_jobs(&e->jobs, &e->g, k_weights, e->full_sums ? 1 : 0, 0);
    tm_make_jobs(&e->jobs_f, &e->g, k_weights, e->full_sums ? 1 : 0, 1);
    const TmJobs &jf = e->jobs_f;
    int tiles = 0, bands = 0;
    for (int k = jf.nfull; k < jf.n; ++k) {
        const TmScaleGeom &sg = e->g.s[jf.scale[k]];
        tiles = std::max(tiles, (sg.w + 31) / 32); bands = std::max(bands, (sg.h + 31) / 32);
    }
    const int ne = jf.n - jf.nfull;
    // a table without EDGE jobs (full_sums, or no SSIMULACRA2) keeps whatever the buffers hold: use_fused_edge looks at the table
    // (jobs_f.n == jobs_f.nfull), and the sizes below stay those of the allocations (resetting ef_ne to 0 here made the
    // next toggle free "0 bytes" and allocate again: mem_usage grew by the buffers' size with every set_full_sums pair)
    if (ne == 0 || !(e->mask & TM_METRIC_SSIMULACRA2)) return TM_OK;
    if (ne > e->ef_ne || tiles > e->ef_tiles || bands > e->ef_bands) {
        if (e->HS) { (void)hipFree(e->HS); e->mem_bytes -= e->hs_elems * sizeof(unsigned long long); e->HS = nullptr; e->hs_elems = 0; }
        if (e->EROWS) { (void)hipFree(e->EROWS); e->mem_bytes -= e->erows_elems * sizeof(double); e->EROWS = nullptr; e->erows_elems = 0; }
        const int ne2 = std::max(ne, e->ef_ne), tiles2 = std::max(tiles, e->ef_tiles), bands2 = std::max(bands, e->ef_bands);
        e->ef_ne = 0; // nothing usable until both allocations exist (use_fused_edge checks it)
        const size_t hs = (size_t)e->cap * ne2 * 2 * tiles2 * 384, er = (size_t)e->cap * ne2 * bands2 * 128;
        int rc;
        if ((rc = dev_alloc(e, &e->HS, hs, true))) { e->HS = nullptr; return rc; }
        e->hs_elems = hs;
        if ((rc = dev_alloc(e, &e->EROWS, er, true))) { e->EROWS = nullptr; return rc; }
        e->erows_elems = er;
        e->ef_ne = ne2; e->ef_tiles = tiles2; e->ef_bands = bands2; // only now: both buffers are there
    }
    if (!e->d_epoch) {
        int rc;
        if ((rc = dev_alloc(e, &e->d_epoch, 2, true))) return rc; // [0] launch epoch of the hand-off tags, [1] ticket counter of the launch (k_finish_edge: epoch + 1, tickets from 0)
        if ((rc = dev_alloc(e, &e->d_status, 8, true))) return rc;
        const unsigned one = 1u;
        HIPCHK(hipMemcpy(e->d_epoch, &one, sizeof one, hipMemcpyHostToDevice));
        e->ef_epoch_host = 1;
        HIPCHK(hipHostMalloc((void **)&e->h_status, 8 * sizeof(int), hipHostMallocDefault));
        *e->h_status = 0;
    }
    return TM_OK;
}

// does a launch of n slots send its EDGE jobs through k_blur_edge_fused?
bool use_fused_edge(const tm_engine *e, int n)
{
    if ((e->variant & (TM_VARIANT_REFERENCE | TM_VARIANT_TWO_PASS_EDGE)) || e->ef_ne == 0 || !e->HS || !e->EROWS || e->jobs_f.n == e->jobs_f.nfull) return false;
    if (e->variant & TM_VARIANT_FUSED_EDGE) return true;
    long long walks = 0;
    for (int k = e->jobs_f.nfull; k < e->jobs_f.n; ++k) walks += (e->g.s[e->jobs_f.scale[k]].h + 31) / 32;
    return walks * n >= e->fused_edge_from;
}

} // namespace

extern "C" {

const char *tm_version(void) { return "turbo-metrics-hip 0.1 (gfx950)"; }

const char *tm_last_hip_error(void) { return g_hip_err; }

const char *tm_strerror(int code)
{
    switch (code) {
    case TM_OK: return "ok";
    case TM_ERR_INVALID_ARG: return "invalid argument";
    case TM_ERR_UNSUPPORTED: return "unsupported (todo!() in the reference: full-range YUV / non-BT.709 transfer, or no gfx950 device)";
    case TM_ERR_HIP: return "HIP runtime error (see tm_last_hip_error)";
    case TM_ERR_OOM: return "out of device memory";
    case TM_ERR_STATE: return "invalid state (compute before all frames set, or results not ready)";
    default: return "unknown error";
    }
}

void *tm_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void tm_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int tm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int tm_device_numa_node(int device)
{
    int n = 0, node = -1;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return -1; }
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return node;
}

int tm_device_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return TM_OK;
}

int tm_init(int device)
{
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { snprintf(g_hip_err, sizeof g_hip_err, "device %d of %d", device, n); return TM_ERR_INVALID_ARG; }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_hip_err, sizeof g_hip_err, "device %d is %s, this library carries gfx950 code only", device, prop.gcnArchName);
        return TM_ERR_UNSUPPORTED;
    }
    return TM_OK;
}

#define TM_V_SLACK ((size_t)4 << 20) /* bytes allocated beyond the pass-1 arena so that its start can be moved */
static std::atomic<int> g_placement_candidates{-1}; // -1: not set -> environment or default

void tm_set_debug_log(int on) { g_debug_log.store(on); }

void tm_set_placement_candidates(int n) { g_placement_candidates.store(n < 1 ? 1 : n); }

// see include/turbo_metrics_hip.h (tm_set_placement_candidates): keep the fastest of a few allocations of the pass-1 arena
static int placement_search(tm_engine *e)
{
    int want = g_placement_candidates.load();
    if (want < 0) { const char *s = getenv("TM_PLACEMENT_CANDIDATES"); want = s ? atoi(s) : 8; }
    const size_t count = (size_t)e->cap * 5 * e->g.pyr_t + TM_V_SLACK / sizeof(float), bytes = count * sizeof(float);
    if (want <= 1 || bytes < ((size_t)1 << 30)) return TM_OK;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return hip_fail(hipGetLastError(), "hipEventCreate");
    // 1. every candidate is allocated first (they all stay alive, so that each lands somewhere else) ...
    std::vector<float *> cand{e->V_alloc};
    for (int t = 1; t < want; ++t) {
        size_t free_b = 0, total_b = 0;
        // memory budget: the candidates alive at once never hold more than a third of the device's memory, and with the next one
        // allocated a quarter of the device must still be free (other tenants of the GPU are not pushed out of memory)
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + total_b / 4 || (size_t)(t + 1) * bytes > total_b / 3) break;
        float *p = nullptr;
        e->V_alloc = nullptr;
        if (dev_alloc(e, &e->V_alloc, count, true) != TM_OK) { (void)hipGetLastError(); break; }
        p = e->V_alloc;
        cand.push_back(p);
    }
    int rc = TM_OK;
    // both kernels that touch the arena: the column pass writes it, the row pass reads it, and they do not always agree on a
    // placement (row pass 1.89 ... 1.98 ms per 64 pairs across candidates) -- the sum decides
    // (with the fused EDGE kernel the two passes run the FULL jobs only: time them as they will run)
    const TmJobs &jb = use_fused_edge(e, (int)e->cap) ? e->jobs_f : e->jobs;
    const dim3 pvgrid((unsigned)e->cap, (unsigned)jb.vstart[jb.nfull], 1), phgrid((unsigned)e->cap, (unsigned)jb.hstart[jb.nfull], 1);
    auto run_once = [&](float *v, float &ms) -> bool {
        (void)hipEventRecord(e0, e->stream);
        hipLaunchKernelGGL((tmk::k_blur_v_jobs<32, 16, 1>), pvgrid, dim3(320), 0, e->stream, e->g, jb, e->XYB, v);
        if (e->g.s[0].w > 2560)
            hipLaunchKernelGGL((tmk::k_blur_h_jobs_x<16, 8, 16, 8, 1>), phgrid, dim3(64), 0, e->stream, e->g, jb, e->XYB, v, e->PART);
        else
            hipLaunchKernelGGL((tmk::k_blur_h_jobs_x<16, 8, 32, 16, 1>), phgrid, dim3(64), 0, e->stream, e->g, jb, e->XYB, v, e->PART);
        (void)hipEventRecord(e1, e->stream);
        return hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    };
    // 2. ... then the device is brought to its steady clock (the first ~100 ms after idle run 10-15 % slower, more than the
    // placements differ: timing the candidates one after the other from a cold start would simply prefer the later ones) ...
    float spent = 0.0f;
    while (rc == TM_OK && cand.size() > 1 && spent < 120.0f) {
        float m = 0.0f;
        if (!run_once(cand[0], m)) rc = hip_fail(hipGetLastError(), "placement search");
        spent += m > 0.0f ? m : 1.0f;
    }
    // 3. ... and the candidates are timed in turns, three rounds, the fastest run of each counts
    std::vector<float> best_of(cand.size(), 1e30f);
    for (int round = 0; rc == TM_OK && cand.size() > 1 && round < 3; ++round)
        for (size_t i = 0; i < cand.size(); ++i) {
            float m = 0.0f;
            if (!run_once(cand[i], m)) { rc = hip_fail(hipGetLastError(), "placement search"); break; }
            if (m < best_of[i]) best_of[i] = m;
        }
    size_t win = 0;
    for (size_t i = 1; i < cand.size(); ++i)
        if (best_of[i] < best_of[win]) win = i;
    if (g_debug_log.load()) // tools/pmc_placement.sh: which candidate is which in the counters
        for (size_t i = 0; i < cand.size(); ++i) fprintf(stderr, "[tm] candidate %zu at %p: %.3f ms%s\n", i, (void *)cand[i], best_of[i], i == win ? " <- kept" : "");
    for (size_t i = 0; i < cand.size(); ++i)
        if (i != win) { (void)hipFree(cand[i]); e->mem_bytes -= bytes; }
    e->V_alloc = cand[win];
    e->V = e->V_alloc;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc == TM_OK && hipMemsetAsync(e->V, 0, bytes, e->stream) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipMemset");
    if (rc == TM_OK && hipStreamSynchronize(e->stream) != hipSuccess) rc = hip_fail(hipGetLastError(), "placement search");
    return rc;
}

int tm_engine_create(tm_engine **out, uint32_t width, uint32_t height, uint32_t metrics_mask, uint32_t batch_capacity)
{
    if (!out) return TM_ERR_INVALID_ARG;
    *out = nullptr;
    if (width == 0 || height == 0 || width > 16384 || height > 16384 || batch_capacity == 0 || batch_capacity > 4096)
        return TM_ERR_INVALID_ARG;
    if ((metrics_mask & ~15u) || metrics_mask == 0) return TM_ERR_INVALID_ARG;
    // SSIM needs one 11x11 window, MS-SSIM one at the fifth dyadic scale
    if ((metrics_mask & TM_METRIC_SSIM) && (width < TM_SSIM_TAPS || height < TM_SSIM_TAPS)) {
        snprintf(g_hip_err, sizeof g_hip_err, "SSIM needs an image of at least 11x11");
        return TM_ERR_UNSUPPORTED;
    }
    if ((metrics_mask & TM_METRIC_MSSSIM) && ((width >> 4) < TM_SSIM_TAPS || (height >> 4) < TM_SSIM_TAPS)) {
        snprintf(g_hip_err, sizeof g_hip_err, "MS-SSIM (5 scales, 11x11 window) needs an image of at least 176x176");
        return TM_ERR_UNSUPPORTED;
    }
    tm_engine *e = new (std::nothrow) tm_engine();
    if (!e) return TM_ERR_OOM;
    int rc = TM_OK;
    auto fail = [&](int code) { tm_engine_destroy(e); return code; };
    if (hipGetDevice(&e->device) != hipSuccess) return fail(hip_fail(hipGetLastError(), "hipGetDevice"));
    e->w = width; e->h = height; e->mask = metrics_mask; e->cap = batch_capacity;
    tm_make_geom(&e->g, (int)width, (int)height);
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cus = prop.multiProcessorCount; }
    // the kernels' code object is loaded HERE, not by the first launch (the reference loads its PTX modules in Ssimulacra2::new:
    // cuModuleLoadData, ssimulacra2-cuda/src/kernel.rs): asking for a kernel's attributes makes the runtime load the module of this library
    { hipFuncAttributes fa; if (hipFuncGetAttributes(&fa, (const void *)tmk::k_finish_jobs) != hipSuccess) (void)hipGetLastError(); }
    hipError_t he = hipSuccess;
    // (from six pairs of 1080p per launch on the edge-only planes take the fused kernel, i.e. the side stream: use_fused_edge)
    if (!(e->stream = engine_stream_acquire(e->device, (unsigned long long)batch_capacity * height < 6ull * 1080ull, &e->lane))) return fail(hip_fail(hipErrorOutOfMemory, "hipStreamCreate"));
    if (!(e->stream2 = side_stream_acquire(e->device))) return fail(hip_fail(hipErrorOutOfMemory, "hipStreamCreate (side stream)"));
    if (!(e->up_stream = up_stream_acquire(e->device))) return fail(hip_fail(hipErrorOutOfMemory, "hipStreamCreate (upload stream)"));
    if ((he = hipEventCreateWithFlags(&e->ev_up_join, hipEventDisableTiming)) != hipSuccess || (he = hipEventCreateWithFlags(&e->ev_stage_free, hipEventDisableTiming)) != hipSuccess) return fail(hip_fail(he, "hipEventCreate"));
    if ((he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming)) != hipSuccess || (he = hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming)) != hipSuccess) return fail(hip_fail(he, "hipEventCreate"));
    if ((he = hipEventCreateWithFlags(&e->ev_col_done, hipEventDisableTiming)) != hipSuccess || (he = hipEventCreateWithFlags(&e->ev_row_done, hipEventDisableTiming)) != hipSuccess) return fail(hip_fail(he, "hipEventCreate"));
    const size_t B = batch_capacity;
    const TmGeom &g = e->g;
    if (metrics_mask & TM_METRIC_SSIMULACRA2) { // PSNR / SSIM / MS-SSIM alone need none of the XYB machinery
        if ((rc = dev_alloc(e, &e->XYB, B * 2 * g.pyr, true))) return fail(rc);
        if ((rc = dev_alloc(e, &e->LIN2, B * 2 * 3 * g.s[2].plane, true))) return fail(rc);
        // LIN and XYBT (linear pyramid, transposed XYB copy) belong to the reference pipeline: allocated by tm_engine_set_variant
        if ((rc = dev_alloc(e, &e->V_alloc, B * 5 * g.pyr_t + TM_V_SLACK / sizeof(float), true))) return fail(rc);
        e->V = e->V_alloc;
        if ((rc = dev_alloc(e, &e->PART, B * 3 * (size_t)g.hblk[TM_SCALES] * 6, true))) return fail(rc);
        if ((rc = dev_alloc(e, &e->SUMS, B * 108, true))) return fail(rc);
    }
    if ((rc = make_job_tables(e))) return fail(rc);
    if ((rc = dev_alloc(e, &e->SSE, B * TM_SSE_BINS * 3, true))) return fail(rc);
    if (metrics_mask & (TM_METRIC_SSIM | TM_METRIC_MSSSIM)) {
        float gw[TM_SSIM_TAPS];
        tm_ssim_window(gw);
        tm_make_ssim_geom(&e->sg, (int)width, (int)height, gw);
        if ((rc = dev_alloc(e, &e->QU8, B * 2 * 3 * e->sg.qplane, true))) return fail(rc);
        if ((rc = dev_alloc(e, &e->SPYR, B * 2 * 3 * e->sg.pyr, true))) return fail(rc);
        if ((rc = dev_alloc(e, &e->SPART, B * 3 * (size_t)e->sg.item_off[TM_SSIM_SCALES] * 2, true))) return fail(rc);
        if ((rc = dev_alloc(e, &e->SSUMS, B * 30, true))) return fail(rc);
        if ((he = hipHostMalloc((void **)&e->h_ssums, B * 30 * sizeof(double), hipHostMallocDefault)) != hipSuccess) return fail(hip_fail(he, "hipHostMalloc"));
    }
    if ((rc = dev_alloc(e, &e->d_desc, B * 2, true))) return fail(rc);
    if ((rc = dev_alloc(e, &e->d_lut, 256, false))) return fail(rc);
    if ((rc = dev_alloc(e, &e->d_coef, 3 * 2 * 5, false))) return fail(rc);
    if ((rc = dev_alloc(e, &e->d_powtab, TM_TAB_DOUBLES, false))) return fail(rc);
    if ((he = hipMemcpy(e->d_powtab, &k_powtab, sizeof k_powtab, hipMemcpyHostToDevice)) != hipSuccess) return fail(hip_fail(he, "hipMemcpy(powtab)"));
    float coef[3][2][5];
    for (int m = 0; m < 3; ++m) { yuv_coefficients(m, 8, coef[m][0]); yuv_coefficients(m, 16, coef[m][1]); }
    if ((he = hipMemcpy(e->d_coef, coef, sizeof coef, hipMemcpyHostToDevice)) != hipSuccess) return fail(hip_fail(he, "hipMemcpy(coef)"));
    if ((he = hipMemcpy(e->d_lut, k_lut_bits, sizeof k_lut_bits, hipMemcpyHostToDevice)) != hipSuccess) return fail(hip_fail(he, "hipMemcpy(lut)"));
    if (e->V && (rc = placement_search(e))) return fail(rc);
    if ((he = hipHostMalloc((void **)&e->h_desc, B * 2 * sizeof(TmFrameDesc), hipHostMallocDefault)) != hipSuccess) return fail(hip_fail(he, "hipHostMalloc"));
    if ((he = hipHostMalloc((void **)&e->h_sums, B * 108 * sizeof(double), hipHostMallocDefault)) != hipSuccess) return fail(hip_fail(he, "hipHostMalloc"));
    if ((he = hipHostMalloc((void **)&e->h_sse, B * TM_SSE_BINS * 3 * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess) return fail(hip_fail(he, "hipHostMalloc"));
    for (size_t i = 0; i < B * 2; ++i) { e->h_desc[i] = TmFrameDesc{nullptr, nullptr, nullptr, 0, 0, TM_KIND_NONE, 0, 0, 0}; }
    e->staging.assign(B * 2, nullptr);
    e->staging_size.assign(B * 2, 0);
    e->staging_own.assign(B * 2, 0);
    for (int i = 0; i < 7; ++i)
        if ((he = hipEventCreate(&e->ev[i])) != hipSuccess) return fail(hip_fail(he, "hipEventCreate"));
    {
        std::lock_guard<std::mutex> lock(g_side_mutex);
        g_engines.push_back(e);
    }
    *out = e;
    return TM_OK;
}

void tm_engine_destroy(tm_engine *e)
{
    if (!e) return;
    {
        std::lock_guard<std::mutex> lock(g_side_mutex);
        for (size_t i = 0; i < g_engines.size();) {
            if (g_engines[i] == e) { g_engines[i] = g_engines.back(); g_engines.pop_back(); continue; }
            if (g_engines[i]->chain_peer == e) g_engines[i]->chain_peer = nullptr; // (its launches would wait on this engine's destroyed events)
            ++i;
        }
    }
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->stream2) (void)hipStreamSynchronize(e->stream2);
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream); // (copies into the staging surfaces that no launch has waited for)
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    for (size_t i = 0; i < e->staging.size(); ++i) if (e->staging[i] && e->staging_own[i]) (void)hipFree(e->staging[i]);
    for (char *a : e->stage_arena) if (a) (void)hipFree(a);
    (void)hipFree(e->LIN); (void)hipFree(e->LIN2); (void)hipFree(e->XYB); (void)hipFree(e->XYBT); (void)hipFree(e->V_alloc);
    (void)hipFree(e->QU8); (void)hipFree(e->SPYR); (void)hipFree(e->SPART); (void)hipFree(e->SSUMS);
    if (e->h_ssums) (void)hipHostFree(e->h_ssums);
    (void)hipFree(e->PART); (void)hipFree(e->SUMS); (void)hipFree(e->SSE); (void)hipFree(e->d_desc);
    (void)hipFree(e->d_lut); (void)hipFree(e->d_coef); (void)hipFree(e->d_powtab);
    (void)hipFree(e->HS); (void)hipFree(e->EROWS); (void)hipFree(e->d_epoch); (void)hipFree(e->d_status);
    if (e->h_status) (void)hipHostFree(e->h_status);
    if (e->h_desc) (void)hipHostFree(e->h_desc);
    if (e->h_sums) (void)hipHostFree(e->h_sums);
    if (e->h_sse) (void)hipHostFree(e->h_sse);
    for (int i = 0; i < 7; ++i) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    for (hipEvent_t ev : e->up_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->up_ev2) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_up_join) (void)hipEventDestroy(e->ev_up_join);
    if (e->ev_stage_free) (void)hipEventDestroy(e->ev_stage_free);
    if (e->up_stream) up_stream_release(e->device);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->ev_col_done) (void)hipEventDestroy(e->ev_col_done);
    if (e->ev_row_done) (void)hipEventDestroy(e->ev_row_done);
    if (e->stream2) side_stream_release(e->device);
    if (e->stream) engine_stream_release(e->device, e->stream, e->lane);
    delete e;
}

size_t tm_engine_mem_usage(const tm_engine *e) { return e ? e->mem_bytes : 0; }

int tm_engine_set_frame_nv12(tm_engine *e, uint32_t slot, int side, const void *y, const void *uv, size_t pitch,
                             int matrix, int transfer, int full_range, int mem)
{
    int rc = check_yuv_args(matrix, transfer, full_range);
    if (rc) return rc;
    return set_frame_common(e, slot, side, TM_KIND_NV12, y, uv, pitch, matrix, mem);
}

int tm_engine_set_frame_p016(tm_engine *e, uint32_t slot, int side, const void *y, const void *uv, size_t pitch,
                             int matrix, int transfer, int full_range, int mem)
{
    int rc = check_yuv_args(matrix, transfer, full_range);
    if (rc) return rc;
    return set_frame_common(e, slot, side, TM_KIND_P016, y, uv, pitch, matrix, mem);
}

// a decoded surface as the reference's decoder hands it over (NvDecNV12 / NvDecP016::from_mapping, cudarse-video/src/dec.rs:299-393):
// ONE allocation, coded_height luma rows at `pitch`, then the interleaved CbCr rows at the same pitch
static int set_surface(tm_engine *e, uint32_t slot, int side, int kind, const void *base, size_t pitch, uint32_t coded_height,
                       int matrix, int transfer, int full_range, int mem)
{
    int rc = check_yuv_args(matrix, transfer, full_range);
    if (rc) return rc;
    if (!e || !base || coded_height < e->h || coded_height > 65536) return TM_ERR_INVALID_ARG;
    return set_frame_common(e, slot, side, kind, base, (const char *)base + pitch * (size_t)coded_height, pitch, matrix, mem, coded_height);
}

int tm_engine_set_surface_nv12(tm_engine *e, uint32_t slot, int side, const void *base, size_t pitch, uint32_t coded_height,
                               int matrix, int transfer, int full_range, int mem)
{
    return set_surface(e, slot, side, TM_KIND_NV12, base, pitch, coded_height, matrix, transfer, full_range, mem);
}

int tm_engine_set_surface_p016(tm_engine *e, uint32_t slot, int side, const void *base, size_t pitch, uint32_t coded_height,
                               int matrix, int transfer, int full_range, int mem)
{
    return set_surface(e, slot, side, TM_KIND_P016, base, pitch, coded_height, matrix, transfer, full_range, mem);
}

int tm_engine_set_frame_i420(tm_engine *e, uint32_t slot, int side, const void *y, const void *u, const void *v, size_t pitch_y,
                             size_t pitch_uv, int bits, int matrix, int transfer, int full_range, int mem)
{
    int rc = check_yuv_args(matrix, transfer, full_range);
    if (rc) return rc;
    return set_frame_planar(e, slot, side, y, u, v, pitch_y, pitch_uv, bits, matrix, mem);
}

int tm_engine_set_frame_i420p10(tm_engine *e, uint32_t slot, int side, const void *y, const void *u, const void *v, size_t pitch_y,
                                size_t pitch_uv, int matrix, int transfer, int full_range, int mem)
{
    int rc = check_yuv_args(matrix, transfer, full_range);
    if (rc) return rc;
    return set_frame_planar(e, slot, side, y, u, v, pitch_y, pitch_uv, 10, matrix, mem, true);
}

size_t tm_p10_row_bytes(uint32_t n_samples) { return (size_t)tm_p10_row_words(n_samples) * 4; }

// one row: word k of block b = s[384 b + k] | s[384 b + 128 + k] << 10 | s[384 b + 256 + k] << 20 (tm_geom.h); three contiguous runs in, one run
// out: the loops below are what a vectorising compiler wants (no gather, no cross-lane step)
static inline __attribute__((always_inline)) void p10_pack_row_body(const uint16_t *__restrict__ s, uint32_t n, uint32_t *__restrict__ d)
{
    uint32_t b = 0;
    for (; (b + 1) * TM_P10_BLOCK <= n; ++b) { // whole blocks
        const uint16_t *__restrict__ s0 = s + (size_t)b * TM_P10_BLOCK, *__restrict__ s1 = s0 + TM_P10_RUN, *__restrict__ s2 = s1 + TM_P10_RUN;
        uint32_t *__restrict__ o = d + (size_t)b * TM_P10_RUN;
        for (int k = 0; k < TM_P10_RUN; ++k) o[k] = ((uint32_t)s0[k] & 1023u) | (((uint32_t)s1[k] & 1023u) << 10) | (((uint32_t)s2[k] & 1023u) << 20);
    }
    if (b * TM_P10_BLOCK < n) { // the last, partial block: absent samples are 0
        uint32_t *o = d + (size_t)b * TM_P10_RUN;
        for (uint32_t k = 0; k < TM_P10_RUN; ++k) {
            uint32_t w = 0;
            for (uint32_t j = 0; j < 3; ++j) {
                const uint32_t x = b * TM_P10_BLOCK + j * TM_P10_RUN + k;
                if (x < n) w |= ((uint32_t)s[x] & 1023u) << (10 * j);
            }
            o[k] = w;
        }
    }
}

// the same loops compiled for the baseline x86-64 (SSE2: four words per step) and for AVX2 (eight), chosen once per process
static void p10_pack_row_sse2(const uint16_t *s, uint32_t n, uint32_t *d) { p10_pack_row_body(s, n, d); }
__attribute__((target("avx2"))) static void p10_pack_row_avx2(const uint16_t *s, uint32_t n, uint32_t *d) { p10_pack_row_body(s, n, d); }

void tm_p10_pack_rows(const void *src, size_t src_pitch, uint32_t width, uint32_t rows, void *dst, size_t dst_pitch)
{
    static void (*const pack)(const uint16_t *, uint32_t, uint32_t *) = __builtin_cpu_supports("avx2") ? p10_pack_row_avx2 : p10_pack_row_sse2;
    for (uint32_t r = 0; r < rows; ++r)
        pack((const uint16_t *)((const char *)src + (size_t)r * src_pitch), width, (uint32_t *)((char *)dst + (size_t)r * dst_pitch));
}

int tm_engine_set_frame_rgb8(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem)
{
    return set_frame_common(e, slot, side, TM_KIND_RGB8, rgb, nullptr, pitch, 0, mem);
}
int tm_engine_set_frame_rgb16(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem)
{
    return set_frame_common(e, slot, side, TM_KIND_RGB16, rgb, nullptr, pitch, 0, mem);
}
int tm_engine_set_frame_rgbf32(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem)
{
    return set_frame_common(e, slot, side, TM_KIND_RGBF32, rgb, nullptr, pitch, 0, mem);
}
int tm_engine_set_frame_linear_f32(tm_engine *e, uint32_t slot, int side, const void *rgb, size_t pitch, int mem)
{
    return set_frame_common(e, slot, side, TM_KIND_LINEARF32, rgb, nullptr, pitch, 0, mem);
}

// ---- upload fences: when may a page-locked host frame be overwritten? --------------------------------------------------------
// A TM_MEM_HOST_PINNED frame is pulled by an asynchronous DMA into an engine-owned device surface; the host bytes are free again
// when THAT copy is done, long before the batch it belongs to has been computed.  A fence marks "every upload enqueued so far".
#define TM_UPLOAD_FENCES 256
int tm_engine_upload_fence(tm_engine *e, uint64_t *token)
{
    if (!e || !token) return TM_ERR_INVALID_ARG;
    TM_BIND(e);
    if (e->up_ev.empty()) { // all or nothing: a ring with holes would let tm_engine_upload_done answer early
        std::vector<hipEvent_t> a(TM_UPLOAD_FENCES, nullptr), b(TM_UPLOAD_FENCES, nullptr);
        hipError_t he = hipSuccess;
        for (size_t k = 0; k < 2 * (size_t)TM_UPLOAD_FENCES && he == hipSuccess; ++k)
            he = hipEventCreateWithFlags(k < TM_UPLOAD_FENCES ? &a[k] : &b[k - TM_UPLOAD_FENCES], hipEventDisableTiming);
        if (he != hipSuccess) {
            for (hipEvent_t ev : a) if (ev) (void)hipEventDestroy(ev);
            for (hipEvent_t ev : b) if (ev) (void)hipEventDestroy(ev);
            return hip_fail(he, "hipEventCreate");
        }
        e->up_ev.swap(a);
        e->up_ev2.swap(b);
        e->up_tok2.assign(TM_UPLOAD_FENCES, UINT64_MAX);
    }
    { const int rc = flush_pending(e); if (rc) return rc; } // a fence covers every upload handed over so far: also the one still held back
    const size_t i = e->up_next % TM_UPLOAD_FENCES;
    HIPCHK(hipEventRecord(e->up_ev[i], e->stream));
    if (e->up_since_fence) { // the copies of this fence that went up on the second upload stream
        HIPCHK(hipEventRecord(e->up_ev2[i], e->up_stream));
        e->up_since_fence = false;
        e->up_last2 = e->up_next;
    }
    // A fence covers EVERY upload enqueued so far: also second-stream copies from before an earlier fence, which nothing on the
    // engine's stream waits for until the next launch -- so each fence remembers the latest second-stream fence.
    e->up_tok2[i] = e->up_last2;
    *token = e->up_next++;
    return TM_OK;
}

// 1: every upload enqueued before the fence has left host memory; 0: not yet (block = 0) -- with block != 0 the call waits for it.
// A token older than the last TM_UPLOAD_FENCES fences is answered by the events that took its place: the slot's event on the engine's
// stream now stands for a later fence of the same in-order stream, and the second upload stream is represented by its most recent
// fence -- done there means done for everything older (conservative, never early).  < 0: error (the negated TM_* code).
int tm_engine_upload_done(tm_engine *e, uint64_t token, int block)
{
    if (!e || token >= e->up_next) return -TM_ERR_INVALID_ARG;
    if (hipSetDevice(e->device) != hipSuccess) { (void)hip_fail(hipGetLastError(), "hipSetDevice"); return -TM_ERR_HIP; }
    const size_t i = token % TM_UPLOAD_FENCES;
    const bool recycled = e->up_next - token > TM_UPLOAD_FENCES;
    hipEvent_t ev2 = nullptr;
    // (the event in the slot of token t was last recorded by a fence >= t of the same in-order stream: done there covers t)
    const uint64_t t2 = recycled ? e->up_last2 : e->up_tok2[i];
    if (t2 != UINT64_MAX) ev2 = e->up_ev2[t2 % TM_UPLOAD_FENCES];
    hipEvent_t evs[2] = {e->up_ev[i], ev2};
    for (hipEvent_t ev : evs) {
        if (!ev) continue; // (no copy ever went up on the second stream)
        if (block) {
            const hipError_t r = hipEventSynchronize(ev);
            if (r != hipSuccess) { (void)hip_fail(r, "hipEventSynchronize"); return -TM_ERR_HIP; }
            continue;
        }
        const hipError_t r = hipEventQuery(ev);
        if (r == hipSuccess) continue;
        if (r == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
        (void)hip_fail(r, "hipEventQuery");
        return -TM_ERR_HIP;
    }
    return 1;
}

int tm_engine_set_profiling(tm_engine *e, int on)
{
    if (!e) return TM_ERR_INVALID_ARG;
    e->profiling = on != 0;
    return TM_OK;
}

int tm_engine_set_variant(tm_engine *e, int variant)
{
    if (!e || (variant & ~(TM_VARIANT_REFERENCE | TM_VARIANT_WIDE_ROWS | TM_VARIANT_TILE_INGEST | TM_VARIANT_SPLIT_ROWS | TM_VARIANT_WHOLE_ROWS | TM_VARIANT_TWO_PASS_EDGE | TM_VARIANT_FUSED_EDGE | TM_VARIANT_UPPER_KERNEL))) return TM_ERR_INVALID_ARG;
    if ((variant & TM_VARIANT_TWO_PASS_EDGE) && (variant & TM_VARIANT_FUSED_EDGE)) return TM_ERR_INVALID_ARG;
    const bool ref = (variant & TM_VARIANT_REFERENCE) != 0;
    // the reference pipeline is SSIMULACRA2 (+ PSNR) only: its ingest kernel neither writes the u8 planes of SSIM / MS-SSIM nor runs without the XYB arenas
    if (ref && ((e->mask & (TM_METRIC_SSIM | TM_METRIC_MSSSIM)) || !(e->mask & TM_METRIC_SSIMULACRA2))) return TM_ERR_INVALID_ARG;
    if (ref && (variant & TM_VARIANT_WIDE_ROWS)) return TM_ERR_INVALID_ARG;
    if (ref && (!e->XYBT || !e->LIN)) { // it keeps the linear pyramid and a transposed XYB copy in HBM
        if (e->in_flight) { int rc = tm_engine_sync(e); if (rc) return rc; }
        int rc = TM_OK;
        if (!e->XYBT) rc = dev_alloc(e, &e->XYBT, (size_t)e->cap * 2 * e->g.pyr_t, true);
        if (rc == TM_OK && !e->LIN) rc = dev_alloc(e, &e->LIN, (size_t)e->cap * 2 * e->g.pyr, true);
        if (rc) return rc;
    }
    e->variant = variant;
    return TM_OK;
}

int tm_engine_set_graph(tm_engine *e, int on)
{
    if (!e) return TM_ERR_INVALID_ARG;
    e->graph_mode = on < 0 ? -1 : on ? 1 : 0; // (negative: back to the default, the engine decides per launch)
    e->use_graph = on > 0;
    e->auto_key = -1; e->auto_seen = 0;
    return TM_OK;
}

int tm_engine_set_full_sums(tm_engine *e, int on)
{
    if (!e) return TM_ERR_INVALID_ARG;
    if (e->in_flight) { int rc = tm_engine_sync(e); if (rc) return rc; }
    TM_BIND(e);
    e->full_sums = on != 0;
    e->have_results = false;
    return make_job_tables(e);
}

int tm_engine_uses_fused_edge(const tm_engine *e, uint32_t n_slots)
{
    if (!e || n_slots == 0 || n_slots > e->cap) return TM_ERR_INVALID_ARG;
    return use_fused_edge(e, (int)n_slots) ? 1 : 0;
}

int tm_engine_get_job_modes(const tm_engine *e, int out[18])
{
    if (!e || !out) return TM_ERR_INVALID_ARG;
    for (int i = 0; i < TM_SCALES * 3; ++i) out[i] = e->jobs.job_of[i] < 0 ? TM_MODE_NONE : e->jobs.mode[e->jobs.job_of[i]];
    return TM_OK;
}

// Launch the whole pipeline for slots [0, n) on stream `st`.  ev (optional): 5 events bracketing the 4 stages.
static int launch_batch(tm_engine *e, hipStream_t st, int n, int want_sse, hipEvent_t *ev)
{
    const TmGeom &g = e->g;
    const TmFrameDesc *h_desc = e->h_desc, *d_desc = e->d_desc;
    const bool ssimu2 = (e->mask & TM_METRIC_SSIMULACRA2) != 0;
    float *XYB = e->XYB, *XYBT = e->XYBT, *V = e->V, *LIN = e->LIN, *LIN2 = e->LIN2;
    (void)XYBT; (void)LIN; // (the reference pipeline's arenas: unused in the ship build)
    double *PART = e->PART, *SUMS = e->SUMS;
    unsigned long long *SSE = e->SSE;
    unsigned char *QU8 = e->QU8;
#ifdef TM_SHIP
    const bool reference = false; // (no way to select it: tm_engine_set_variant is not part of the ship library)
#else
    const bool reference = (e->variant & TM_VARIANT_REFERENCE) != 0;
#endif
    const bool chained = e->chain_peer != nullptr && !e->use_graph;
    if (chained) HIPCHK(hipStreamWaitEvent(st, e->chain_peer->ev_col_done, 0)); // (an event never recorded counts as complete)
    if (ev) HIPCHK(hipEventRecord(ev[0], st));
    // ---- stage INGEST: frames -> linear RGB -> XYB pyramid
    if (reference) { // separate straight-line kernels, linear pyramid in HBM, two plain XYB pyramids [side][scale][channel]
#ifndef TM_SHIP
        const int qw = ((int)e->w + 1) / 2, qh = ((int)e->h + 1) / 2;
        dim3 grid((unsigned)((qw + 63) / 64), (unsigned)((qh + 3) / 4), (unsigned)n), block(64, 4, 1);
        hipLaunchKernelGGL(tmk::k_ingest, grid, block, 0, st, g, d_desc, e->d_lut, e->d_coef, e->d_powtab, LIN, SSE, want_sse);
        for (int s = 1; s < TM_SCALES; ++s)
            hipLaunchKernelGGL(tmk::k_downscale, grid2(g.s[s].w, g.s[s].h, n * 6), dim3(64), 0, st, g, s, LIN);
        for (int s = 0; s < TM_SCALES; ++s)
            hipLaunchKernelGGL(tmk::k_xyb, grid2(g.s[s].w, g.s[s].h, n * 2), dim3(64), 0, st, g, s, LIN, XYB);
#endif
    } else {
        dim3 grid((unsigned)((e->w + 31) / 32), (unsigned)((e->h + 7) / 8), (unsigned)n);
        int kind = h_desc[0].kind; // one format for the whole launch (the normal case) -> specialised kernel
        for (int i = 1; i < 2 * n; ++i) if (h_desc[i].kind != kind) kind = -1;
#define TM_LAUNCH_W(K) hipLaunchKernelGGL((tmk::k_ingest_wave<K>), grid, dim3(64), 0, st, g, d_desc, e->d_lut, e->d_coef, e->d_powtab, XYB, LIN2, SSE, want_sse, QU8, e->sg.qplane, e->sg.pitch[0])
        // the 4:2:0 kinds: side-packed row-walking kernel, one wave = 64 quads x rows_per_wave quad rows (e->ingest_rows; even)
        const int qw = ((int)e->w + 1) / 2, qh = ((int)e->h + 1) / 2;
        int rpw = e->ingest_rows;
        if (rpw <= 0) { // enough waves to fill the chip several times over, as few table stagings as that allows (8 vs 16 rows per
            // wave, 64 1080p pairs: 1.115 vs 1.14 ms; 4: 1.34; 2: 1.38 -- tools/ingest_ab.py)
            rpw = 8;
            while (rpw > 2 && (long long)((qw + 63) / 64) * ((qh + rpw - 1) / rpw) * n < 16384) rpw /= 2;
        }
        const bool rows = !(e->variant & TM_VARIANT_TILE_INGEST);
        const bool yuv420 = kind == TM_KIND_NV12 || kind == TM_KIND_P016 || kind == TM_KIND_I420_8 || kind == TM_KIND_I420_16 || kind == TM_KIND_I420_P10;
        // levels 2..5 in the row-walking kernel's own epilogue (a workgroup = 128 x 16 rpw pixels = whole level-5 pixels for rpw 4 or 8); the
        // tile kernel, other values of the test hook and TM_VARIANT_UPPER_KERNEL go through LIN2 and k_ingest_upper_rd
        const bool fold = ssimu2 && rows && yuv420 && !(e->variant & TM_VARIANT_UPPER_KERNEL) && (e->ingest_rows <= 0 || e->ingest_rows == 4 || e->ingest_rows == 8);
        if (fold && rpw < 4) rpw = 4;
        dim3 rgrid((unsigned)((qw + 63) / 64), (unsigned)((qh + 4 * rpw - 1) / (4 * rpw)), (unsigned)n);
        const tmk::TmIngestGeom ig = tmk::tm_ingest_geom(g);
        const bool quant = want_sse || QU8 != nullptr;
#define TM_LAUNCH_RF(K, Q, F) hipLaunchKernelGGL((tmk::k_ingest_rows<K, Q, F>), rgrid, dim3(256), 0, st, ig, d_desc, e->d_coef, e->d_powtab, XYB, LIN2, SSE, want_sse, QU8, e->sg.qplane, e->sg.pitch[0], rpw)
#define TM_LAUNCH_R(K) do { if (quant) { if (fold) TM_LAUNCH_RF(K, true, true); else TM_LAUNCH_RF(K, true, false); } \
                        else { if (fold) TM_LAUNCH_RF(K, false, true); else TM_LAUNCH_RF(K, false, false); } } while (0)
        switch (kind) {
        case TM_KIND_NV12: if (rows) TM_LAUNCH_R(TM_KIND_NV12); else TM_LAUNCH_W(TM_KIND_NV12); break;
        case TM_KIND_P016: if (rows) TM_LAUNCH_R(TM_KIND_P016); else TM_LAUNCH_W(TM_KIND_P016); break;
        case TM_KIND_I420_8: if (rows) TM_LAUNCH_R(TM_KIND_I420_8); else TM_LAUNCH_W(TM_KIND_I420_8); break;
        case TM_KIND_I420_16: if (rows) TM_LAUNCH_R(TM_KIND_I420_16); else TM_LAUNCH_W(TM_KIND_I420_16); break;
        case TM_KIND_I420_P10: if (rows) TM_LAUNCH_R(TM_KIND_I420_P10); else TM_LAUNCH_W(TM_KIND_I420_P10); break;
        case TM_KIND_RGB8: TM_LAUNCH_W(TM_KIND_RGB8); break;
        case TM_KIND_RGB16: TM_LAUNCH_W(TM_KIND_RGB16); break;
        case TM_KIND_RGBF32: TM_LAUNCH_W(TM_KIND_RGBF32); break;
        case TM_KIND_LINEARF32: TM_LAUNCH_W(TM_KIND_LINEARF32); break;
        default: TM_LAUNCH_W(-1); break;
        }
#undef TM_LAUNCH_R
#undef TM_LAUNCH_RF
#undef TM_LAUNCH_W
        // levels 2..5 (when the ingest kernel has not produced them itself)
        if (ssimu2 && !fold) hipLaunchKernelGGL(tmk::k_ingest_upper_rd, dim3((unsigned)((g.s[2].w + 31) / 32), (unsigned)((g.s[2].h + 31) / 32), (unsigned)n), dim3(256), 0, st, g, LIN2, XYB);
    }
    if (ev) HIPCHK(hipEventRecord(ev[1], st));
    // ---- SSIM / MS-SSIM on the u8 planes the ingest kernel wrote (tm_ssim_kernels.h)
    bool ssim_done = false;
    auto launch_ssim = [&](hipStream_t ss) {
        const TmSsimGeom &sg = e->sg;
        const int nscales = (e->mask & TM_METRIC_MSSSIM) ? TM_SSIM_SCALES : 1;
        if (nscales > 1)
            hipLaunchKernelGGL(tmk::k_ssim_pyramid, dim3((unsigned)((sg.w[0] + 127) / 128), (unsigned)((sg.h[0] + 31) / 32), (unsigned)(n * 6)), dim3(64), 0, ss, sg, QU8, e->SPYR);
        // the sum of l * cs is needed on scale 0 for SSIM and on the last scale for MS-SSIM (the others use cs alone)
        const unsigned need_l = e->full_sums ? 31u : ((e->mask & TM_METRIC_SSIM) ? 1u : 0u) | ((e->mask & TM_METRIC_MSSSIM) ? 1u << (TM_SSIM_SCALES - 1) : 0u);
        hipLaunchKernelGGL(tmk::k_ssim_stream, dim3((unsigned)(n * 3), (unsigned)sg.item_off[nscales], 1), dim3(64), 0, ss, sg, nscales, need_l, QU8, e->SPYR, e->SPART);
        hipLaunchKernelGGL(tmk::k_ssim_finish, dim3((unsigned)n, 30, 1), dim3(64), 0, ss, sg, nscales, e->SPART, e->SSUMS);
        ssim_done = true;
    };
    const bool has_ssim_stage = (e->mask & (TM_METRIC_SSIM | TM_METRIC_MSSSIM)) != 0;
    // (the SSIM stage only needs the u8 planes of the ingest kernel and is bound by arithmetic while the blur passes are bound by
    // HBM -- but running it on a second stream beside them was measured: 7.63 k vs 7.77 k pairs/s, docs/LABBOOK.md section 5.1)
    if (ssimu2) {
        // the EDGE jobs (scale 0 of X and B with the reference's weights) go through ONE kernel without the pass-1 arena; the two
        // passes then run the FULL jobs only (the first nfull entries of the edge-last table)
        const bool fused = use_fused_edge(e, n);
        TmJobs jobs = fused ? e->jobs_f : e->jobs;
        jobs.prio = fused && e->ef_beside > 0 && !e->use_graph ? e->ef_pass_prio : 0;
        const long long hblocks = jobs.hstart[jobs.nfull];
        const dim3 vgrid((unsigned)n, (unsigned)jobs.vstart[jobs.nfull], 1), hgrid((unsigned)n, (unsigned)hblocks, 1);
        // The fused kernel is bound by what the SIMDs can issue and reads 2 of the 14 units of a FULL job; the two passes are bound
        // by HBM: it runs on a second stream beside them (fork behind the ingest stage, join in front of the finisher).  Enqueued
        // first it takes the chip and the column pass moves into the slots its long tail leaves; small launches, where it cannot
        // fill the chip, gain most (8 1080p pairs 0.91 -> 0.85 ms, 64: 5.5 -> 4.7).
        auto launch_fused = [&](hipStream_t fs) -> int {
            const int ne = jobs.n - jobs.nfull, planes = n * ne;
            int bands = 0;
            for (int k = jobs.nfull; k < jobs.n; ++k) bands = std::max(bands, (g.s[jobs.scale[k]].h + 31) / 32);
            tmk::TmEdgeArgs ea;
            tmk::tm_make_edge_args(&ea, &g, &jobs, e->ef_tiles, e->ef_bands);
            const int dbg = e->ef_fault;
            if (ev) HIPCHK(hipEventRecord(ev[5], fs));
            // four waves per workgroup: four adjacent bands of one plane (ef_waves 4, default: the state crosses three of four band
            // boundaries through LDS), or the same band of four planes (ef_waves 5: tuning), or single-wave workgroups (1: tuning)
            const bool grouped = e->ef_waves == 4;
            const int nw = e->ef_waves == 1 ? 1 : 4, groups = grouped ? (bands + 3) / 4 : (planes + nw - 1) / nw;
            const unsigned total = grouped ? (unsigned)planes * (unsigned)groups : (unsigned)groups * (unsigned)bands;
            // beside the two passes the kernel is a PERSISTENT launch of 7/8 of a workgroup per CU (four waves: one per SIMD) that share
            // the tickets: it then never holds more than a share of a CU the passes can work beside (they keep two of three column-pass
            // workgroups / five of eight row-pass waves per CU and, with raised wave priority, the issue slots they need), runs for
            // about as long as they do and hides behind them -- 64 1080p pairs 4.70 ms with one workgroup per ticket, 4.54 so.
            // Alone on the chip (behind the row pass): one workgroup per ticket.
            const unsigned wgs = fs == st || e->ef_persist_wgs < 0 ? total : std::min(total, (unsigned)(e->ef_persist_wgs > 0 ? e->ef_persist_wgs : e->n_cus * 7 / 8));
            if (nw == 1) hipLaunchKernelGGL((tmk::k_blur_edge_fused<1, false>), dim3(wgs), dim3(64), 0, fs, ea, planes, groups, total, XYB, e->HS, e->d_epoch, e->d_epoch + 1, e->EROWS, e->d_status, dbg);
            else if (grouped) hipLaunchKernelGGL((tmk::k_blur_edge_fused<4, true>), dim3(wgs), dim3(256), 0, fs, ea, planes, groups, total, XYB, e->HS, e->d_epoch, e->d_epoch + 1, e->EROWS, e->d_status, dbg);
            else hipLaunchKernelGGL((tmk::k_blur_edge_fused<4, false>), dim3(wgs), dim3(256), 0, fs, ea, planes, groups, total, XYB, e->HS, e->d_epoch, e->d_epoch + 1, e->EROWS, e->d_status, dbg);
            hipLaunchKernelGGL(tmk::k_finish_edge, dim3((unsigned)planes), dim3(64), 0, fs, ea, e->EROWS, PART, e->d_epoch);
            if (ev) HIPCHK(hipEventRecord(ev[6], fs));
            HIPCHK(hipMemcpyAsync(e->h_status, e->d_status, sizeof(int), hipMemcpyDeviceToHost, fs));
            return TM_OK;
        };
        const bool beside = fused && e->ef_beside > 0 && !e->use_graph; // (a captured sequence stays on the engine's own stream: the side stream is shared between engines)
        if (ev) e->edge_timed = fused;
        if (chained) HIPCHK(hipStreamWaitEvent(st, e->chain_peer->ev_row_done, 0));
        if (beside) { HIPCHK(hipEventRecord(e->ev_fork, st)); HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_fork, 0)); }
        if (beside && e->ef_beside == 1) { int rc = launch_fused(e->stream2); if (rc) return rc; }
        // ---- stage BLUR_V: column pass, all scales / channels / slots in one launch
#ifndef TM_SHIP
        if (reference) hipLaunchKernelGGL(tmk::k_blur_v, dim3((unsigned)g.vblk[TM_SCALES], 3, (unsigned)n), dim3(64), 0, st, g, XYB, XYBT, V);
        else
#endif
        // few column blocks (a pair or two per launch): every role-wave as a workgroup of its own, a SIMD each (tm_kernels.h)
        if (vgrid.y && 5ll * n * vgrid.y <= e->solo_col_below) hipLaunchKernelGGL((tmk::k_blur_v_jobs<32, 16, 0, true>), dim3(vgrid.x, vgrid.y, 5), dim3(64), 0, st, g, jobs, XYB, V);
        else if (vgrid.y) hipLaunchKernelGGL((tmk::k_blur_v_jobs<32, 16>), vgrid, dim3(320), 0, st, g, jobs, XYB, V);
        if (ev) HIPCHK(hipEventRecord(ev[2], st));
        if (!e->use_graph) HIPCHK(hipEventRecord(e->ev_col_done, st));
        if (beside && e->ef_beside != 1) { int rc = launch_fused(e->stream2); if (rc) return rc; }
        // ---- stage BLUR_H: row pass + error maps + reductions
#ifndef TM_SHIP
        if (reference) hipLaunchKernelGGL(tmk::k_blur_h_jobs, dim3((unsigned)jobs.hstart[TM_MAX_JOBS], 1, (unsigned)n), dim3(64), 0, st, g, jobs, XYBT, V, PART);
        else
#endif
        if (!hgrid.y) {}
        // few row blocks (small launches): eight waves per block -- what one CU can issue for one row block, not the chip, bounds this pass then
        // (faster than the one-wave pass up to 32 1080p pairs beside the fused kernel = 2 432 row blocks of FULL jobs -- 13.2 k vs 13.0 k pairs/s --,
        // slower at 48 = 3 648 blocks: 13.6 k vs 13.75 k; profiles/r04e_split8_threshold.log)
        else if ((e->variant & TM_VARIANT_SPLIT_ROWS) || (!(e->variant & TM_VARIANT_WHOLE_ROWS) && (long long)n * hblocks <= (beside && !e->split_rows_env ? 2600 : e->split_rows_below))) {
            hipLaunchKernelGGL(tmk::k_blur_h_jobs_split, hgrid, dim3(64 * TM_SPLIT_WAVES), 0, st, g, jobs, XYB, V, PART);
        }
        else if (g.s[0].w > 2560 || (e->variant & TM_VARIANT_WIDE_ROWS)) hipLaunchKernelGGL((tmk::k_blur_h_jobs_x<16, 8, 16, 8>), hgrid, dim3(64), 0, st, g, jobs, XYB, V, PART);
        else hipLaunchKernelGGL((tmk::k_blur_h_jobs_x<16, 8, 32, 16>), hgrid, dim3(64), 0, st, g, jobs, XYB, V, PART);
        if (ev) HIPCHK(hipEventRecord(ev[3], st));
        if (!e->use_graph) HIPCHK(hipEventRecord(e->ev_row_done, st));
        // ---- stage EDGE (when not beside the passes): both recurrences, the edge maps and their sums of the EDGE jobs in one kernel
        if (fused && !beside) { int rc = launch_fused(st); if (rc) return rc; }
        if (beside) { HIPCHK(hipEventRecord(e->ev_join, e->stream2)); HIPCHK(hipStreamWaitEvent(st, e->ev_join, 0)); }
        hipLaunchKernelGGL(tmk::k_finish_jobs, dim3((unsigned)n), dim3(128), 0, st, jobs, PART, SUMS);
    } else if (ev) {
        e->edge_timed = false;
        HIPCHK(hipEventRecord(ev[2], st));
        HIPCHK(hipEventRecord(ev[3], st));
    }
    if (has_ssim_stage && !ssim_done) launch_ssim(st);
    if (ev) HIPCHK(hipEventRecord(ev[4], st));
    return TM_OK;
}

int tm_engine_compute_async(tm_engine *e, uint32_t n_slots)
{
    if (!e || n_slots == 0 || n_slots > e->cap) return TM_ERR_INVALID_ARG;
    TM_BIND(e);
    for (uint32_t i = 0; i < n_slots * 2; ++i)
        if (e->h_desc[i].kind == TM_KIND_NONE) return TM_ERR_STATE;
    if (e->ev_pending) { // fold the previous compute's timings before the events are reused
        int rc = tm_engine_sync(e);
        if (rc) return rc;
    }
    { const int rc = flush_pending(e); if (rc) return rc; }
    hipStream_t st = e->stream;
    const int n = (int)n_slots;
    bool fused_launch = false;
    const int want_sse = (e->mask & TM_METRIC_PSNR) ? 1 : 0;
    if (e->up_pending) { // frames that went up on the second upload stream: the launch waits for them
        HIPCHK(hipEventRecord(e->ev_up_join, e->up_stream));
        HIPCHK(hipStreamWaitEvent(st, e->ev_up_join, 0));
        e->up_pending = false;
    }
    // everything one batch enqueues: descriptor upload, accumulator reset, the kernels, result download
    auto enqueue_all = [&]() -> int {
        HIPCHK(hipMemcpyAsync(e->d_desc, e->h_desc, (size_t)n * 2 * sizeof(TmFrameDesc), hipMemcpyHostToDevice, st));
        if (want_sse) hipLaunchKernelGGL(tmk::k_zero_u64, dim3(((unsigned)n * TM_SSE_BINS * 3 + 255) / 256), dim3(256), 0, st, e->SSE, (unsigned)n * TM_SSE_BINS * 3);
        {
            int rc = launch_batch(e, st, n, want_sse, e->profiling ? e->ev : nullptr);
            if (rc) return rc;
        }
        if (e->mask & TM_METRIC_SSIMULACRA2)
            HIPCHK(hipMemcpyAsync(e->h_sums, e->SUMS, (size_t)n * 108 * sizeof(double), hipMemcpyDeviceToHost, st));
        if (want_sse) HIPCHK(hipMemcpyAsync(e->h_sse, e->SSE, (size_t)n * TM_SSE_BINS * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        if (e->mask & (TM_METRIC_SSIM | TM_METRIC_MSSSIM))
            HIPCHK(hipMemcpyAsync(e->h_ssums, e->SSUMS, (size_t)n * 30 * sizeof(double), hipMemcpyDeviceToHost, st));
        return TM_OK;
    };
    // The sequence is the same from batch to batch (frame pointers travel through h_desc, which the captured copy node
    // re-reads at every replay), so it is captured once into a hipGraph and replayed: one submission instead of ~10.
    // Key = everything the launch code branches on.  Profiling (events between the stages) launches directly.
    if ((e->mask & TM_METRIC_SSIMULACRA2) && use_fused_edge(e, n)) {
        // The hand-off tags carry 24 bits of the launch epoch, and words of planes that recent launches did not touch (fewer slots)
        // keep their old tags: after a wrap an old tag could match a new launch's.  Whenever the epoch is back at 1 --
        // every 2^24 - 1 fused launches -- the words are cleared first (tag 0 is never expected); on the engine's stream, in front of
        // everything this launch enqueues or replays.
        if (e->ef_epoch_unknown) { // an earlier fused launch failed part-way or bailed out: ask the device where its epoch stands
            HIPCHK(hipStreamSynchronize(st));
            if (e->stream2) HIPCHK(hipStreamSynchronize(e->stream2));
            HIPCHK(hipMemcpy(&e->ef_epoch_host, e->d_epoch, sizeof(unsigned), hipMemcpyDeviceToHost));
            // the failed launch may have run k_blur_edge_fused without k_finish_edge: the device epoch then still reads E while words
            // tagged E are already in HS, and the next launch -- tag E again -- would accept them without waiting: clear them
            if (e->HS) HIPCHK(hipMemsetAsync(e->HS, 0, e->hs_elems * sizeof(unsigned long long), st));
        }
        else if (e->ef_epoch_host == 1u && e->HS) HIPCHK(hipMemsetAsync(e->HS, 0, e->hs_elems * sizeof(unsigned long long), st));
        e->ef_epoch_unknown = true; // until this launch has been enqueued in full (any early return below leaves it set)
        fused_launch = true;
    }
    int kind = e->h_desc[0].kind;
    for (int i = 1; i < 2 * n; ++i) if (e->h_desc[i].kind != kind) kind = -1;
    const long long key = ((long long)n << 40) ^ ((long long)(kind + 2) << 32) ^ ((long long)e->variant << 4) ^ (e->full_sums ? 1 : 0);
    // Replay from a captured graph: always (tm_engine_set_graph 1), never (0), or -- the default -- by itself for launches of up to four pairs
    // that run without the fused kernel, once the same launch shape has come four times in a row: one submission instead of ~8 is worth
    // 3-4 % there (1 / 2 / 4 pairs of 1080p: 3 145 / 5 465 / 8 714 against 3 028 / 5 303 / 8 493 pairs/s, profiles/r06z_small_launch_probe.log),
    // while larger launches lose the fused kernel's second stream to it.  A caller whose launches change shape never pays for a capture.
    if (e->graph_mode >= 0) e->use_graph = e->graph_mode == 1;
    else if (!fused_launch && n <= 4 && !e->profiling && !e->chain_peer) {
        if (key == e->auto_key) { if (e->auto_seen < 1000) ++e->auto_seen; } else { e->auto_key = key; e->auto_seen = 1; }
        e->use_graph = e->auto_seen > 3;
    } else { e->auto_key = -1; e->auto_seen = 0; e->use_graph = false; }
    bool launched = false;
    if (e->use_graph && !e->profiling) {
        if (!e->gexec || e->gkey != key) {
            if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = enqueue_all();
                const hipError_t ce = hipStreamEndCapture(st, &graph);
                if (rc == TM_OK && ce == hipSuccess && graph && hipGraphInstantiate(&e->gexec, graph, nullptr, nullptr, 0) == hipSuccess) e->gkey = key;
                else e->gexec = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!e->gexec) { (void)hipGetLastError(); e->use_graph = false; e->graph_mode = 0; } // this runtime cannot capture the sequence: launch directly from now on
        }
        if (e->gexec) {
            HIPCHK(hipGraphLaunch(e->gexec, st));
            launched = true;
        }
    }
    if (!launched) {
        int rc = enqueue_all();
        if (rc) return rc;
    }
    HIPCHK(hipGetLastError());
    if (fused_launch) { // enqueued in full: the mirror takes the step k_finish_edge takes on the device
        const unsigned next = (e->ef_epoch_host + 1u) & 0xFFFFFFu;
        e->ef_epoch_host = next ? next : 1u;
        e->ef_epoch_unknown = false;
    }
    if (e->up_stream && e->upload_streams >= 2) { // the next frame that goes up on the second upload stream must not overtake this launch
        HIPCHK(hipEventRecord(e->ev_stage_free, st));
        e->stage_busy = true;
    }
    e->ev_pending = e->profiling;
    e->last_n = n_slots;
    e->in_flight = true;
    e->have_results = false;
    return TM_OK;
}

int tm_engine_sync(tm_engine *e)
{
    if (!e) return TM_ERR_INVALID_ARG;
    TM_BIND(e);
    { const int rc = flush_pending(e); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->up_pending) { // frames handed over since the last launch: "valid until tm_engine_sync" holds for them too
        HIPCHK(hipStreamSynchronize(e->up_stream));
        e->up_pending = false;
    }
    if (e->ev_pending) {
        // TM_STAGE_INGEST, BLUR_V, BLUR_H, SSIM: consecutive events on the engine's stream (SSIM: finisher + SSIM kernels, and the wait
        // for the fused kernel if it is still running beside); TM_STAGE_EDGE: the fused kernel's own pair, on the stream it ran on
        for (int i = 0; i < TM_STAGE_COUNT; ++i) {
            float ms = 0.0f;
            if (i == TM_STAGE_EDGE) { if (e->edge_timed) HIPCHK(hipEventElapsedTime(&ms, e->ev[5], e->ev[6])); }
            else HIPCHK(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
            e->stage_ms[i] += (double)ms;
        }
        e->n_prof += 1;
        e->ev_pending = false;
    }
    if (e->in_flight) { e->in_flight = false; e->have_results = true; }
    if (e->h_status && *e->h_status) { // k_blur_edge_fused gave up waiting for the band above: the sums of this launch are not valid
        *e->h_status = 0;
        (void)hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream);
        (void)hipStreamSynchronize(e->stream);
        e->have_results = false;
        e->ef_epoch_unknown = true;
        snprintf(g_hip_err, sizeof g_hip_err, "k_blur_edge_fused: a state hand-off between bands timed out");
        return TM_ERR_HIP;
    }
    return TM_OK;
}

int tm_engine_get_stage_ms(tm_engine *e, double ms[TM_STAGE_COUNT], uint64_t *n_computes, int reset)
{
    if (!e || !ms) return TM_ERR_INVALID_ARG;
    for (int i = 0; i < TM_STAGE_COUNT; ++i) ms[i] = e->stage_ms[i];
    if (n_computes) *n_computes = e->n_prof;
    if (reset) { for (int i = 0; i < TM_STAGE_COUNT; ++i) e->stage_ms[i] = 0.0; e->n_prof = 0; }
    return TM_OK;
}

double tm_ssimulacra2_score_from_sums(const double sums[108], uint32_t width, uint32_t height)
{
    // post_process_scores, ssimulacra2-cuda/src/lib.rs:586-622
    double sc[108];
    memcpy(sc, sums, sizeof sc);
    int w = (int)width, h = (int)height;
    for (int scale = 0; scale < TM_SCALES; ++scale) {
        const double opp = 1.0 / (double)(h * w); // NppiRect::norm, cudarse-npp-sys/src/lib.rs:19-21
        for (int c = 0; c < 3; ++c) {
            const int o = 18 * scale + c, ow = c * 36 + 6 * scale;
            sc[o] = std::fabs(sc[o] * opp) * k_weights[ow];
            sc[o + 3] = std::fabs(sc[o + 3] * opp) * k_weights[ow + 1];
            sc[o + 6] = std::fabs(sc[o + 6] * opp) * k_weights[ow + 2];
            sc[o + 9] = std::sqrt(std::sqrt(sc[o + 9] * opp)) * k_weights[ow + 3];
            sc[o + 12] = std::sqrt(std::sqrt(sc[o + 12] * opp)) * k_weights[ow + 4];
            sc[o + 15] = std::sqrt(std::sqrt(sc[o + 15] * opp)) * k_weights[ow + 5];
        }
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    double score = 0.0;
    for (int i = 0; i < 108; ++i) score += sc[i];
    score *= 0.9562382616834844;
    score = std::fma(6.248496625763138e-5 * score * score, score,
                     std::fma(2.326765642916932, score, -0.020884521182843837 * score * score));
    if (score > 0.0) score = std::fma(std::pow(score, 0.6276336467831387), -10.0, 100.0);
    else score = 100.0;
    return score;
}

void tm_ssim_window(float g[11])
{
    // g[k] = exp(-(k-5)^2 / (2 * 1.5^2)) / sum, evaluated in f64 and rounded to f32
    double v[TM_SSIM_TAPS], sum = 0.0;
    for (int k = 0; k < TM_SSIM_TAPS; ++k) { const double d = (double)(k - 5); v[k] = std::exp(-(d * d) / (2.0 * 1.5 * 1.5)); sum += v[k]; }
    for (int k = 0; k < TM_SSIM_TAPS; ++k) g[k] = (float)(v[k] / sum);
}

static double ssim_channel(const double sums[30], uint32_t width, uint32_t height, int c)
{
    return sums[(c * TM_SSIM_SCALES + 0) * 2] / ((double)(width - 10) * (double)(height - 10));
}

static double msssim_channel(const double sums[30], uint32_t width, uint32_t height, int c)
{
    static const double wt[TM_SSIM_SCALES] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};
    double prod = 1.0;
    uint32_t sw = width, sh = height;
    for (int s = 0; s < TM_SSIM_SCALES; ++s) {
        const double n = (double)(sw - 10) * (double)(sh - 10);
        const double v = sums[(c * TM_SSIM_SCALES + s) * 2 + (s == TM_SSIM_SCALES - 1 ? 0 : 1)] / n;
        prod *= std::pow(v > 0.0 ? v : 0.0, wt[s]);
        sw /= 2; sh /= 2;
    }
    return prod;
}

double tm_ssim_channel_from_sums(const double sums[30], uint32_t width, uint32_t height, int channel)
{
    if (width < TM_SSIM_TAPS || height < TM_SSIM_TAPS || channel < 0 || channel > 2) return NAN;
    return (double)(float)ssim_channel(sums, width, height, channel);
}

double tm_msssim_channel_from_sums(const double sums[30], uint32_t width, uint32_t height, int channel)
{
    if ((width >> 4) < TM_SSIM_TAPS || (height >> 4) < TM_SSIM_TAPS || channel < 0 || channel > 2) return NAN;
    return (double)(float)msssim_channel(sums, width, height, channel);
}

double tm_ssim_from_sums(const double sums[30], uint32_t width, uint32_t height)
{
    if (width < TM_SSIM_TAPS || height < TM_SSIM_TAPS) return NAN;
    double acc = 0.0;
    for (int c = 0; c < 3; ++c) acc += ssim_channel(sums, width, height, c);
    return (double)(float)(acc / 3.0); // one Npp32f read back (ist.rs:118,133), widened (lib.rs:355-357)
}

double tm_msssim_from_sums(const double sums[30], uint32_t width, uint32_t height)
{
    if ((width >> 4) < TM_SSIM_TAPS || (height >> 4) < TM_SSIM_TAPS) return NAN;
    double acc = 0.0;
    for (int c = 0; c < 3; ++c) acc += msssim_channel(sums, width, height, c);
    return (double)(float)(acc / 3.0);
}

int tm_engine_get_ssim_sums(tm_engine *e, uint32_t slot, double out[30])
{
    if (!e || !out || slot >= e->cap) return TM_ERR_INVALID_ARG;
    if (!e->have_results || slot >= e->last_n || !(e->mask & (TM_METRIC_SSIM | TM_METRIC_MSSSIM))) return TM_ERR_STATE;
    memcpy(out, e->h_ssums + (size_t)slot * 30, 30 * sizeof(double));
    return TM_OK;
}

int tm_engine_get_raw_sums(tm_engine *e, uint32_t slot, double out[108])
{
    if (!e || !out || slot >= e->cap) return TM_ERR_INVALID_ARG;
    if (!e->have_results || slot >= e->last_n || !(e->mask & TM_METRIC_SSIMULACRA2)) return TM_ERR_STATE;
    memcpy(out, e->h_sums + (size_t)slot * 108, 108 * sizeof(double));
    return TM_OK;
}

int tm_engine_get_sse(tm_engine *e, uint32_t slot, uint64_t *out)
{
    if (!e || !out || slot >= e->cap) return TM_ERR_INVALID_ARG;
    if (!e->have_results || slot >= e->last_n || !(e->mask & TM_METRIC_PSNR)) return TM_ERR_STATE;
    *out = slot_sse(e, slot);
    return TM_OK;
}

int tm_engine_get_sse_channels(tm_engine *e, uint32_t slot, uint64_t out[3])
{
    if (!e || !out || slot >= e->cap) return TM_ERR_INVALID_ARG;
    if (!e->have_results || slot >= e->last_n || !(e->mask & TM_METRIC_PSNR)) return TM_ERR_STATE;
    for (int c = 0; c < 3; ++c) out[c] = slot_sse_channel(e, slot, c);
    return TM_OK;
}

int tm_engine_set_channel_mode(tm_engine *e, int mode)
{
    if (!e || (mode != TM_CHANNELS_POOLED && mode != TM_CHANNELS_FIRST)) return TM_ERR_INVALID_ARG;
    e->channel_mode = mode;
    return TM_OK;
}

double tm_psnr_from_sse(uint64_t sse, uint64_t n_samples)
{
    const double mse = (double)sse / (double)n_samples;
    return (double)(float)(10.0 * std::log10(255.0 * 255.0 / mse)); // one Npp32f (ist.rs:118), widened (lib.rs:355)
}

int tm_engine_get_scores(tm_engine *e, uint32_t slot, tm_frame_scores *out)
{
    if (!e || !out || slot >= e->cap) return TM_ERR_INVALID_ARG;
    if (!e->have_results || slot >= e->last_n) return TM_ERR_STATE;
    memset(out, 0, sizeof *out);
    if (e->mask & TM_METRIC_SSIMULACRA2) {
        out->ssimulacra2 = tm_ssimulacra2_score_from_sums(e->h_sums + (size_t)slot * 108, e->w, e->h);
        out->valid |= TM_METRIC_SSIMULACRA2;
    }
    if (e->mask & TM_METRIC_PSNR) {
        // PSNR of the u8-quantised linear RGB pair (turbo-metrics/src/lib.rs:296-318); NPP returns one
        // Npp32f (cudarse-npp/src/image/ist.rs:118) which the engine widens (lib.rs:355).
        const uint64_t px = (uint64_t)e->w * e->h;
        out->psnr = e->channel_mode == TM_CHANNELS_FIRST ? tm_psnr_from_sse(slot_sse_channel(e, slot, 0), px)
                                                         : tm_psnr_from_sse(slot_sse(e, slot), 3 * px);
        out->valid |= TM_METRIC_PSNR;
    }
    if (e->mask & TM_METRIC_SSIM) {
        out->ssim = e->channel_mode == TM_CHANNELS_FIRST ? tm_ssim_channel_from_sums(e->h_ssums + (size_t)slot * 30, e->w, e->h, 0)
                                                         : tm_ssim_from_sums(e->h_ssums + (size_t)slot * 30, e->w, e->h);
        out->valid |= TM_METRIC_SSIM;
    }
    if (e->mask & TM_METRIC_MSSSIM) {
        out->msssim = e->channel_mode == TM_CHANNELS_FIRST ? tm_msssim_channel_from_sums(e->h_ssums + (size_t)slot * 30, e->w, e->h, 0)
                                                           : tm_msssim_from_sums(e->h_ssums + (size_t)slot * 30, e->w, e->h);
        out->valid |= TM_METRIC_MSSSIM;
    }
    return TM_OK;
}

int tm_engine_get_scores_batch(tm_engine *e, uint32_t first_slot, uint32_t n, tm_frame_scores *out)
{
    if (!e || !out || first_slot >= e->cap || n > e->cap - first_slot) return TM_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        const int rc = tm_engine_get_scores(e, first_slot + i, out + i);
        if (rc) return rc;
    }
    return TM_OK;
}

int tm_engine_debug_set_v_offset(tm_engine *e, size_t bytes)
{
    if (!e || !e->V_alloc || bytes > TM_V_SLACK || (bytes & 15)) return TM_ERR_INVALID_ARG;
    if (e->in_flight) { int rc = tm_engine_sync(e); if (rc) return rc; }
    e->V = e->V_alloc + bytes / sizeof(float);
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; e->gkey = -1; } // captured launches hold the old pointer
    return TM_OK;
}

int tm_engine_debug_set_edge_beside(tm_engine *e, int mode)
{
    if (!e || mode < 0 || mode > 2) return TM_ERR_INVALID_ARG;
    if (e->in_flight) { int rc = tm_engine_sync(e); if (rc) return rc; }
    e->ef_beside = mode;
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; e->gkey = -1; }
    return TM_OK;
}

int tm_engine_debug_set_edge_epoch(tm_engine *e, uint32_t epoch)
{
    if (!e || !e->d_epoch || epoch == 0 || epoch > 0xFFFFFFu) return TM_ERR_INVALID_ARG;
    TM_BIND(e);
    if (e->in_flight) { int rc = tm_engine_sync(e); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->d_epoch, &epoch, sizeof epoch, hipMemcpyHostToDevice));
    e->ef_epoch_host = epoch;
    e->ef_epoch_unknown = false;
    return TM_OK;
}

int tm_engine_debug_set_param(tm_engine *e, int param, long long value)
{
    if (!e) return TM_ERR_INVALID_ARG;
    if (e->in_flight) { int rc = tm_engine_sync(e); if (rc && param != TM_DBG_EF_FAULT) return rc; }
    if (param == TM_DBG_UPLOAD_STREAMS || param == TM_DBG_UPLOAD_MERGE) { TM_BIND(e); const int rc = flush_pending(e); if (rc) return rc; }
    switch (param) {
    case TM_DBG_FUSED_EDGE_FROM: if (value < 0) return TM_ERR_INVALID_ARG; e->fused_edge_from = value; break;
    case TM_DBG_EF_WAVES: if (value != 1 && value != 4 && value != 5) return TM_ERR_INVALID_ARG; e->ef_waves = (int)value; break;
    case TM_DBG_EF_PERSIST_WGS: if (value < -1 || value > 65536) return TM_ERR_INVALID_ARG; e->ef_persist_wgs = (int)value; break;
    case TM_DBG_PASS_PRIO: if (value < 0 || value > 1) return TM_ERR_INVALID_ARG; e->ef_pass_prio = (int)value; break;
    case TM_DBG_SPLIT_ROWS_BELOW: if (value < 0) return TM_ERR_INVALID_ARG; e->split_rows_below = value; e->split_rows_env = true; break;
    case TM_DBG_EF_FAULT: if (value < 0 || value > 3) return TM_ERR_INVALID_ARG; e->ef_fault = (int)value; break;
    case TM_DBG_SOLO_COL_BELOW: if (value < 0) return TM_ERR_INVALID_ARG; e->solo_col_below = value; break;
    case TM_DBG_UPLOAD_STREAMS: if (value < 1 || value > 2) return TM_ERR_INVALID_ARG; if (e->up_pending) { HIPCHK(hipStreamSynchronize(e->up_stream)); e->up_pending = false; } e->upload_streams = (int)value; break;
    case TM_DBG_LINEAR_UPLOAD: if (value < 0 || value > 1) return TM_ERR_INVALID_ARG; e->dbg_no_linear_upload = value ? 0 : 1; break;
    case TM_DBG_UPLOAD_MERGE: if (value < 0 || value > ((long long)1 << 32)) return TM_ERR_INVALID_ARG; e->merge_limit = (size_t)value; break;
    default: return TM_ERR_INVALID_ARG;
    }
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; e->gkey = -1; } // captured launches hold the old values
    return TM_OK;
}

int tm_engine_set_linear_upload(tm_engine *e, int on)
{
    return tm_engine_debug_set_param(e, TM_DBG_LINEAR_UPLOAD, on ? 1 : 0);
}

int tm_engine_debug_chain(tm_engine *e, tm_engine *peer)
{
    if (!e || peer == e || (peer && peer->device != e->device)) return TM_ERR_INVALID_ARG;
    if (e->in_flight) { int rc = tm_engine_sync(e); if (rc) return rc; }
    e->chain_peer = peer;
    return TM_OK;
}

int tm_engine_debug_set_ingest_rows(tm_engine *e, int rows)
{
    if (!e || rows < 0 || rows > 128 || (rows & 1)) return TM_ERR_INVALID_ARG;
    if (e->in_flight) { int rc = tm_engine_sync(e); if (rc) return rc; }
    e->ingest_rows = rows;
    if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; e->gkey = -1; }
    return TM_OK;
}

int tm_engine_debug_read_plane(tm_engine *e, uint32_t slot, int kind, int scale, int index, int channel, float *out,
                               size_t out_count)
{
    if (!e || !out || slot >= e->cap || scale < 0 || scale >= TM_SCALES || channel < 0 || channel > 2) return TM_ERR_INVALID_ARG;
    const TmGeom &g = e->g;
    const TmScaleGeom &sg = g.s[scale];
    if (out_count < (size_t)sg.w * sg.h) return TM_ERR_INVALID_ARG;
    if (!(e->mask & TM_METRIC_SSIMULACRA2)) return TM_ERR_STATE; // no XYB / pass-1 planes without SSIMULACRA2
    TM_BIND(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    const float *src = nullptr;
    size_t pitch = 0, width = 0, rows = 0;
    switch (kind) {
    case TM_PLANE_LINEAR:
    case TM_PLANE_XYB:
        if (index < 0 || index > 1) return TM_ERR_INVALID_ARG;
        if (kind == TM_PLANE_XYB && !(e->variant & TM_VARIANT_REFERENCE)) { // ref/dis-interleaved pyramid: de-interleave on the host
            std::vector<float> rows((size_t)sg.h * sg.pitch * 2);
            HIPCHK(hipMemcpy(rows.data(), e->XYB + (size_t)slot * 2 * g.pyr + 2 * (sg.off + channel * sg.plane), rows.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (int y = 0; y < sg.h; ++y)
                for (int x = 0; x < sg.w; ++x) out[(size_t)y * sg.w + x] = rows[2 * ((size_t)y * sg.pitch + x) + index];
            return TM_OK;
        }
        if (kind == TM_PLANE_LINEAR && (!e->LIN || !(e->variant & TM_VARIANT_REFERENCE))) return TM_ERR_STATE; // only the reference pipeline stores it
        src = (kind == TM_PLANE_LINEAR ? e->LIN : e->XYB) + (size_t)(slot * 2 + index) * g.pyr + sg.off + channel * sg.plane;
        pitch = sg.pitch; width = sg.w; rows = sg.h;
        break;
    case TM_PLANE_XYB_T:
        if (index < 0 || index > 1) return TM_ERR_INVALID_ARG;
        if (!(e->variant & TM_VARIANT_REFERENCE) || !e->XYBT) return TM_ERR_STATE; // the default pipeline writes no transposed copy
        src = e->XYBT + (size_t)(slot * 2 + index) * g.pyr_t + sg.off_t + channel * sg.plane_t;
        pitch = sg.pitch_t; width = sg.h; rows = sg.w;
        break;
    case TM_PLANE_PASS1_T:
        if (index < 0 || index > 4) return TM_ERR_INVALID_ARG;
        src = e->V + (size_t)(slot * 5 + index) * g.pyr_t + sg.off_t + channel * sg.plane_t;
        pitch = sg.pitch_t; width = sg.h; rows = sg.w;
        break;
    default: return TM_ERR_INVALID_ARG;
    }
    HIPCHK(hipMemcpy2D(out, width * sizeof(float), src, pitch * sizeof(float), width * sizeof(float), rows, hipMemcpyDeviceToHost));
    return TM_OK;
}

} // extern "C"
