// turbo_metrics.cpp -- see turbo_metrics.hpp.  Host orchestration only: frame selection, batching over the engine's
// slots, ping-pong pipelining of two engines; every number comes out of libturbometrics_hip.so.
#include "turbo_metrics.hpp"
#include <dlfcn.h>
#include <chrono>
#include <fstream>
#include <sched.h>
#include <string>

#include <condition_variable>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace tm_host {

TmError::TmError(int c, const std::string &where)
    : std::runtime_error(where + ": " + tm_strerror(c) + (c == TM_ERR_HIP ? std::string(" [") + tm_last_hip_error() + "]" : std::string())),
      code(c)
{
}

static void chk(int code, const char *where)
{
    if (code != TM_OK) throw TmError(code, where);
}

// The calling thread (and every thread it starts afterwards: the sources' readers, the ring helper, the fetch helper) is bound to the CPUs
// of the NUMA node next to the device.  Frames are copied from the page cache into page-locked memory that the runtime places on that
// node and are then read by the device's copy engines through that socket: from the other socket the same CLI run delivers 4.3-4.9 k
// pairs/s of 1080p instead of 6.6-7.4 k, and left to the scheduler it lands in between and differs from box to box
// (tools/numa_probe.sh, profiles/r04y_numa_probe.log).  TM_NUMA_BIND=0 leaves the affinity alone; nothing happens when the node is
// unknown or the process may not run on any of its CPUs.
static void bind_to_device_node(int device)
{
    const char *env = getenv("TM_NUMA_BIND");
    if (env && atoi(env) == 0) return;
    const int node = tm_device_numa_node(device);
    if (node < 0) return;
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string list;
    if (!(f >> list)) return;
    // The mask the PROCESS was started with, taken once before the first bind: with --devices N the main thread is bound to device 0's
    // node before the shard threads start, they inherit that narrowed mask, and a device on the other socket would intersect to nothing
    // and stay on device 0's CPUs.
    static cpu_set_t original;
    static std::once_flag once;
    static bool have_original = false;
    std::call_once(once, [] { have_original = sched_getaffinity(getpid(), sizeof original, &original) == 0; });
    if (!have_original) return;
    cpu_set_t now = original, want;
    CPU_ZERO(&want);
    int n = 0;
    for (size_t i = 0; i < list.size();) { // "0-63,128-191"
        char *end = nullptr;
        const long lo = strtol(list.c_str() + i, &end, 10);
        long hi = lo;
        if (end && *end == '-') hi = strtol(end + 1, &end, 10);
        if (!end || lo < 0 || hi < lo) return;
        for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET((int)c, &now)) { CPU_SET((int)c, &want); ++n; }
        i = (size_t)(end - list.c_str());
        if (i < list.size() && list[i] == ',') ++i; else break;
    }
    if (n > 0 && n < CPU_COUNT(&now)) (void)sched_setaffinity(0, sizeof want, &want);
}

void init_hip(int device)
{
    chk(tm_init(device), "tm_init");
    bind_to_device_node(device);
}

// ---- colour metadata ---------------------------------------------------------------------------------------------
ColorCharacteristics ColorCharacteristics::from_codes(int cp, int mc, int tc)
{
    ColorCharacteristics c;
    switch (cp) {
    case 1: c.cp = ColourPrimaries::BT709; break;
    case 2: c.cp = ColourPrimaries::Unspecified; break;
    case 5: c.cp = ColourPrimaries::BT601_625; break;
    case 6: c.cp = ColourPrimaries::BT601_525; break;
    case 0: case 3: c.cp = ColourPrimaries::Invalid; break;
    default: c.cp = ColourPrimaries::Unsupported; break;
    }
    switch (mc) {
    case 1: c.mc = MatrixCoefficients::BT709; break;
    case 2: c.mc = MatrixCoefficients::Unspecified; break;
    case 5: c.mc = MatrixCoefficients::BT601_625; break;
    case 6: c.mc = MatrixCoefficients::BT601_525; break;
    case 3: c.mc = MatrixCoefficients::Invalid; break;
    default: c.mc = MatrixCoefficients::Unsupported; break; // 0 = identity, 4 = FCC, ...
    }
    switch (tc) {
    case 1: case 6: c.tc = TransferCharacteristic::BT709; break;
    case 2: c.tc = TransferCharacteristic::Unspecified; break;
    case 0: case 3: c.tc = TransferCharacteristic::Invalid; break;
    default: c.tc = TransferCharacteristic::Unsupported; break;
    }
    return c;
}

ColorCharacteristics ColorCharacteristics::or_(const ColorCharacteristics &o) const
{
    ColorCharacteristics r = *this;
    if (cp == ColourPrimaries::Unspecified || cp == ColourPrimaries::Invalid) r.cp = o.cp;
    if (mc == MatrixCoefficients::Unspecified || mc == MatrixCoefficients::Invalid) r.mc = o.mc;
    if (tc == TransferCharacteristic::Unspecified || tc == TransferCharacteristic::Invalid) r.tc = o.tc;
    return r;
}

ColorCharacteristics color_characteristics_fallback(uint32_t height)
{
    ColorCharacteristics c;
    c.tc = TransferCharacteristic::BT709;
    if (height <= 525) { c.cp = ColourPrimaries::BT601_525; c.mc = MatrixCoefficients::BT601_525; }
    else if (height <= 625) { c.cp = ColourPrimaries::BT601_625; c.mc = MatrixCoefficients::BT601_625; }
    else { c.cp = ColourPrimaries::BT709; c.mc = MatrixCoefficients::BT709; }
    return c;
}

int get_color_matrix(const ColorCharacteristics &c)
{
    if (c.cp == ColourPrimaries::BT709 && c.mc == MatrixCoefficients::BT709) return TM_MATRIX_BT709;
    if (c.cp == ColourPrimaries::BT601_525 && c.mc == MatrixCoefficients::BT601_525) return TM_MATRIX_BT601_525;
    if (c.cp == ColourPrimaries::BT601_625 && c.mc == MatrixCoefficients::BT601_625) return TM_MATRIX_BT601_625;
    throw std::runtime_error(std::string("not implemented: colour primaries ") + to_string(c.cp) + " with matrix coefficients " +
                             to_string(c.mc) + " (todo!() in the reference, turbo-metrics/src/color.rs:85)");
}

int get_transfer(const ColorCharacteristics &c)
{
    if (c.tc == TransferCharacteristic::BT709) return TM_TRANSFER_BT709;
    throw std::runtime_error(std::string("not implemented: transfer characteristic ") + to_string(c.tc) +
                             " (todo!() in the reference, turbo-metrics/src/color.rs:92)");
}

const char *to_string(ColourPrimaries v)
{
    switch (v) {
    case ColourPrimaries::Invalid: return "Invalid";
    case ColourPrimaries::Unspecified: return "Unspecified";
    case ColourPrimaries::Unsupported: return "Unsupported";
    case ColourPrimaries::BT709: return "BT709";
    case ColourPrimaries::BT601_525: return "BT601_525";
    default: return "BT601_625";
    }
}
const char *to_string(MatrixCoefficients v)
{
    switch (v) {
    case MatrixCoefficients::Invalid: return "Invalid";
    case MatrixCoefficients::Unspecified: return "Unspecified";
    case MatrixCoefficients::Unsupported: return "Unsupported";
    case MatrixCoefficients::BT709: return "BT709";
    case MatrixCoefficients::BT601_525: return "BT601_525";
    default: return "BT601_625";
    }
}
const char *to_string(TransferCharacteristic v)
{
    switch (v) {
    case TransferCharacteristic::Invalid: return "Invalid";
    case TransferCharacteristic::Unspecified: return "Unspecified";
    case TransferCharacteristic::Unsupported: return "Unsupported";
    default: return "BT709";
    }
}
const char *to_string(ColorRange v) { return v == ColorRange::Full ? "Full" : "Limited"; }

// ---- engine --------------------------------------------------------------------------------------------------------
TurboMetrics::TurboMetrics(uint32_t width, uint32_t height, const Metrics &metrics, uint32_t batch, bool pipeline)
    : w_(width), h_(height), batch_(batch ? batch : 1), metrics_(metrics)
{
    chk(tm_engine_create(&eng_[0], w_, h_, metrics_.mask(), batch_), "tm_engine_create");
    if (pipeline) {
        const int rc = tm_engine_create(&eng_[1], w_, h_, metrics_.mask(), batch_);
        if (rc != TM_OK) {
            tm_engine_destroy(eng_[0]);
            eng_[0] = nullptr;
            throw TmError(rc, "tm_engine_create (second engine of the pipeline)");
        }
    }
    // compute_all hands pictures over one by one, as they arrive, with a fence per pair: for that pattern a picture of a planar file is
    // fastest as one linear copy (the library's default suits callers that queue whole batches; tm_engine.hip, set_frame_planar)
    for (tm_engine *e : eng_)
        if (e) (void)tm_engine_set_linear_upload(e, 1);
}

TurboMetrics::~TurboMetrics()
{
    for (tm_engine *e : eng_)
        if (e) tm_engine_destroy(e);
}

size_t TurboMetrics::mem_usage() const
{
    size_t total = 0;
    for (tm_engine *e : eng_)
        if (e) total += tm_engine_mem_usage(e);
    return total;
}

// tm_engine_debug_set_param is a function of the LABORATORY build (include/turbo_metrics_hip_debug.h): the ship library this file links
// does not export it.  A measurement run puts the laboratory build in front (LD_PRELOAD=turbo-metrics_amd/lab/libturbometrics_hip_lab.so,
// tools/cli_ab.sh); the CLI's --tune says so otherwise.
using debug_set_param_fn = int (*)(tm_engine *, int, long long);
static debug_set_param_fn lab_set_param()
{
    static const debug_set_param_fn fn = (debug_set_param_fn)dlsym(RTLD_DEFAULT, "tm_engine_debug_set_param");
    return fn;
}

void TurboMetrics::debug_set_param(int param, long long value)
{
    if (!lab_set_param()) throw std::runtime_error("tuning values belong to the laboratory build: run with LD_PRELOAD=<...>/turbo-metrics_amd/lab/libturbometrics_hip_lab.so");
    retire_deferred(); // a setting never changes under a pair in flight: its scores are collected first
    debug_params_.emplace_back(param, value);
    for (tm_engine *e : eng_)
        if (e) chk(lab_set_param()(e, param, value), "tm_engine_debug_set_param");
}

void TurboMetrics::set_full_sums(bool on)
{
    retire_deferred(); // tm_engine_set_full_sums drops the engine's results: a pair in flight keeps its scores for collect()
    full_sums_ = on; // (an engine created later -- compute_one_deferred's second one -- starts with it)
    for (tm_engine *e : eng_)
        if (e) chk(tm_engine_set_full_sums(e, on ? 1 : 0), "tm_engine_set_full_sums");
}

// == convert_frame_to_linearrgb (color.rs:96-116): kernel selection by frame kind and colour metadata
void TurboMetrics::set_frame(tm_engine *e, uint32_t slot, int side, const HwFrame &f, const ColorInfo &c)
{
    const int mem = f.device ? TM_MEM_DEVICE : (f.pinned ? TM_MEM_HOST_PINNED : TM_MEM_HOST);
    switch (f.kind) {
    case HwFrame::NvDecNV12:
    case HwFrame::NvDecP016: {
        const int matrix = get_color_matrix(c.first), transfer = get_transfer(c.first);
        const int full = c.second == ColorRange::Full ? 1 : 0;
        if (f.kind == HwFrame::NvDecNV12)
            chk(tm_engine_set_frame_nv12(e, slot, side, f.data, f.uv, f.pitch, matrix, transfer, full, mem), "tm_engine_set_frame_nv12");
        else
            chk(tm_engine_set_frame_p016(e, slot, side, f.data, f.uv, f.pitch, matrix, transfer, full, mem), "tm_engine_set_frame_p016");
        break;
    }
    case HwFrame::Planar420: {
        const int matrix = get_color_matrix(c.first), transfer = get_transfer(c.first);
        const int full = c.second == ColorRange::Full ? 1 : 0;
        chk(tm_engine_set_frame_i420(e, slot, side, f.data, f.u, f.v, f.pitch, f.pitch_uv, f.bits, matrix, transfer, full, mem), "tm_engine_set_frame_i420");
        break;
    }
    case HwFrame::Planar420P10: {
        const int matrix = get_color_matrix(c.first), transfer = get_transfer(c.first);
        const int full = c.second == ColorRange::Full ? 1 : 0;
        chk(tm_engine_set_frame_i420p10(e, slot, side, f.data, f.u, f.v, f.pitch, f.pitch_uv, matrix, transfer, full, mem), "tm_engine_set_frame_i420p10");
        break;
    }
    case HwFrame::Npp8: chk(tm_engine_set_frame_rgb8(e, slot, side, f.data, f.pitch, mem), "tm_engine_set_frame_rgb8"); break;
    case HwFrame::Npp16: chk(tm_engine_set_frame_rgb16(e, slot, side, f.data, f.pitch, mem), "tm_engine_set_frame_rgb16"); break;
    case HwFrame::Npp32: chk(tm_engine_set_frame_rgbf32(e, slot, side, f.data, f.pitch, mem), "tm_engine_set_frame_rgbf32"); break;
    }
}

FrameScores TurboMetrics::scores_of(tm_engine *e, uint32_t slot)
{
    tm_frame_scores s;
    chk(tm_engine_get_scores(e, slot, &s), "tm_engine_get_scores");
    FrameScores r;
    if (s.valid & TM_METRIC_PSNR) r.psnr = s.psnr;
    if (s.valid & TM_METRIC_SSIM) r.ssim = s.ssim;
    if (s.valid & TM_METRIC_MSSSIM) r.msssim = s.msssim;
    if (s.valid & TM_METRIC_SSIMULACRA2) r.ssimulacra2 = s.ssimulacra2;
    return r;
}

void TurboMetrics::create_deferred_engine(size_t i)
{
    if (eng_[i]) return;
    chk(tm_engine_create(&eng_[i], w_, h_, metrics_.mask(), 1), "tm_engine_create (further engine of compute_one_deferred)");
    (void)tm_engine_set_linear_upload(eng_[i], 1);
    if (full_sums_) chk(tm_engine_set_full_sums(eng_[i], 1), "tm_engine_set_full_sums");
    for (const auto &kv : debug_params_) (void)lab_set_param()(eng_[i], kv.first, kv.second); // (only ever non-empty with the laboratory build loaded)
}

void TurboMetrics::set_deferred_depth(uint32_t depth, bool create_now)
{
    if (depth < 2 || depth > MAX_DEFERRED_DEPTH) throw TmError(TM_ERR_INVALID_ARG, "set_deferred_depth: 2 ... 8 pairs in flight");
    retire_deferred();
    for (size_t i = std::max<size_t>(2, depth); i < eng_.size(); ++i)
        if (eng_[i]) tm_engine_destroy(eng_[i]);
    eng_.resize(std::max<size_t>(2, depth), nullptr);
    def_pending_.assign(depth, 0);
    def_depth_ = depth;
    if (create_now) {
        if (batch_ != 1) throw TmError(TM_ERR_INVALID_ARG, "compute_one_deferred: create the TurboMetrics object with batch = 1");
        for (size_t i = 1; i < depth; ++i) create_deferred_engine(i);
    }
}

void TurboMetrics::retire_deferred()
{
    for (size_t i = 0; i < def_pending_.size(); ++i)
        if (def_pending_[i]) {
            chk(tm_engine_sync(eng_[i]), "tm_engine_sync");
            def_done_.emplace_back(def_pending_[i], scores_of(eng_[i], 0));
            def_pending_[i] = 0;
        }
}

FrameScores TurboMetrics::compute_one(const HwFrame &fref, const ColorInfo &cref, const HwFrame &fdis, const ColorInfo &cdis)
{
    retire_deferred(); // (slot 0 of this engine may hold a deferred pair)
    set_frame(eng_[0], 0, TM_SIDE_REF, fref, cref);
    set_frame(eng_[0], 0, TM_SIDE_DIS, fdis, cdis);
    chk(tm_engine_compute_async(eng_[0], 1), "tm_engine_compute_async");
    chk(tm_engine_sync(eng_[0]), "tm_engine_sync");
    return scores_of(eng_[0], 0);
}

uint64_t TurboMetrics::compute_one_deferred(const HwFrame &fref, const ColorInfo &cref, const HwFrame &fdis, const ColorInfo &cdis)
{
    // the one-pair-per-call path, as in the Python mirror: on a batched object the lazily created second engine would cost a whole
    // batch of device memory for one-pair launches
    if (batch_ != 1) throw TmError(TM_ERR_INVALID_ARG, "compute_one_deferred: create the TurboMetrics object with batch = 1");
    const uint64_t ticket = def_next_++;
    const size_t i = (size_t)(ticket % def_depth_);
    create_deferred_engine(i); // (the engines the launches take turns on are created when their turn first comes, unless set_deferred_depth made them)
    if (def_pending_[i]) { // `depth` pairs are in flight already: the oldest one (on this engine) is finished first
        chk(tm_engine_sync(eng_[i]), "tm_engine_sync");
        def_done_.emplace_back(def_pending_[i], scores_of(eng_[i], 0));
        def_pending_[i] = 0;
    }
    set_frame(eng_[i], 0, TM_SIDE_REF, fref, cref);
    set_frame(eng_[i], 0, TM_SIDE_DIS, fdis, cdis);
    chk(tm_engine_compute_async(eng_[i], 1), "tm_engine_compute_async");
    def_pending_[i] = ticket;
    return ticket;
}

FrameScores TurboMetrics::collect(uint64_t ticket)
{
    for (size_t i = 0; i < def_pending_.size(); ++i)
        if (ticket && def_pending_[i] == ticket) {
            def_pending_[i] = 0;
            chk(tm_engine_sync(eng_[i]), "tm_engine_sync");
            return scores_of(eng_[i], 0);
        }
    for (size_t k = 0; k < def_done_.size(); ++k)
        if (def_done_[k].first == ticket) {
            FrameScores r = def_done_[k].second;
            def_done_.erase(def_done_.begin() + (long)k);
            return r;
        }
    throw TmError(TM_ERR_INVALID_ARG, "collect: no such ticket (never issued, or collected already)");
}

MetricsResults aggregate_scores(const std::vector<FrameScores> &frames, const Metrics &metrics)
{
    std::optional<std::vector<double>> a, b, c, d;
    if (metrics.psnr) a.emplace();
    if (metrics.ssim) b.emplace();
    if (metrics.msssim) c.emplace();
    if (metrics.ssimulacra2) d.emplace();
    for (const FrameScores &r : frames) {
        if (a && r.psnr) a->push_back(*r.psnr);
        if (b && r.ssim) b->push_back(*r.ssim);
        if (c && r.msssim) c->push_back(*r.msssim);
        if (d && r.ssimulacra2) d->push_back(*r.ssimulacra2);
    }
    MetricsResults res;
    res.frame_count = frames.size();
    if (frames.empty() && (a || b || c || d))
        throw NoFramesSelected();
    if (a) res.psnr = MetricAggregate::from(std::move(*a));
    if (b) res.ssim = MetricAggregate::from(std::move(*b));
    if (c) res.msssim = MetricAggregate::from(std::move(*c));
    if (d) res.ssimulacra2 = MetricAggregate::from(std::move(*d));
    return res;
}

// A page-locked frame is pulled into an engine-owned device surface by an asynchronous DMA; its bytes are free again when THAT
// copy is done (tm_engine_upload_fence / tm_engine_upload_done), not when its batch has been computed: a frame has to survive
// only UPLOADS_IN_FLIGHT further next_frame calls, whatever the batch size -- the sources' rings of page-locked surfaces stay
// small (round 3: 2 * batch + 1 surfaces per stream; page-locking them, at ~3 GB/s, was most of a short 4K run and the reason
// why --batch 16 was slower than --batch 8)
// (Up to 16 pairs in flight with a fence per 4 was measured too -- TurboMetrics::set_upload_tuning, `--tune 100=16 --tune 101=4`: the loop
// then waits for the readers instead of the uploads and ends within the run-to-run spread of the default, 7.06 k vs 6.69 k pairs/s on one
// box, 5.87 k vs 6.04 k on another; profiles/r04y_cli_ab2.log, r04y_cli_ab3.log -- the small ring kept.)
// (Round 6: the engine sends page-locked pictures that lie back to back in the source's ring up as ONE DMA, up to 14 MB of them, and every fence
// flushes what it holds back: pictures of up to 1080p 8-bit get a fence per 4 pairs with 8 pairs in flight -- 7.84-7.92 k -> 8.08-8.21 k pairs/s,
// profiles/r06o_cli_fence_ab.log --; larger ones are a DMA each and keep the small ring.)
static size_t UPLOADS_IN_FLIGHT = 4; // pairs whose uploads may be in flight behind the one being read (the sources' lookahead)
static size_t FENCE_EVERY = 1;       // pairs per fence
static bool UPLOAD_TUNING_SET = false;
void TurboMetrics::set_upload_tuning(size_t in_flight, size_t fence_every)
{
    FENCE_EVERY = std::max<size_t>(1, fence_every);
    UPLOADS_IN_FLIGHT = std::max(std::max<size_t>(in_flight ? 1 : 4, in_flight), FENCE_EVERY);
    UPLOAD_TUNING_SET = true;
}

void TurboMetrics::prepare_sources(FrameSource &frames_ref, FrameSource &frames_dis, const Options &opts, size_t min_lookahead)
{
    if (!UPLOAD_TUNING_SET) {
        const bool small = (size_t)frames_ref.width() * frames_ref.height() <= (size_t)1920 * 1088 * 2; // pictures of which several fit one DMA
        UPLOADS_IN_FLIGHT = small ? 8 : 4;
        FENCE_EVERY = small ? 4 : 1;
    }
    for (FrameSource *s : {&frames_ref, &frames_dis}) {
        s->set_lookahead(std::max(UPLOADS_IN_FLIGHT, min_lookahead));
        s->set_readahead(opts.every <= 1); // dropped pictures are consumed without being read: no reading ahead then
        s->prepare();
    }
}

MetricsResults TurboMetrics::compute_all(FrameSource &frames_ref, FrameSource &frames_dis, const Options &opts,
                                         const std::function<void(const FrameScores &)> &on_frame, uint32_t *decode_count_out)
{
    if (frames_ref.width() != frames_dis.width() || frames_ref.height() != frames_dis.height())
        throw std::runtime_error("Reference and distorted are not the same size"); // assert_eq! at lib.rs:368-372
    const ColorInfo cref = frames_ref.color_characteristics(), cdis = frames_dis.color_characteristics();
    retire_deferred(); // (the engines' slots may hold deferred pairs)

    std::optional<std::vector<double>> s_psnr, s_ssim, s_msssim, s_ssimu;
    if (metrics_.psnr) s_psnr.emplace();
    if (metrics_.ssim) s_ssim.emplace();
    if (metrics_.msssim) s_msssim.emplace();
    if (metrics_.ssimulacra2) s_ssimu.emplace();

    uint32_t decode_count = opts.decode_start;
    size_t compute_count = 0;
    timing_ = LoopTiming{};
    const auto tick = [] { return std::chrono::steady_clock::now(); };
    const auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    prepare_sources(frames_ref, frames_dis, opts); // (no-ops when the caller has done it already)
    struct Fence { tm_engine *e = nullptr; uint64_t token = 0; };
    std::vector<Fence> fences(UPLOADS_IN_FLIGHT + FENCE_EVERY + 2); // [pair % size]: the fence that covers the pair's uploads
    size_t first_unfenced = 0;                                      // pairs [first_unfenced, kept) have no fence yet (fewer than FENCE_EVERY, all on eng_[cur])
    size_t kept = 0; // pairs handed to an engine so far
    frames_ref.skip_frames(opts.skip_ref + opts.skip + opts.decode_start);
    frames_dis.skip_frames(opts.skip_dis + opts.skip + opts.decode_start);

    // A batch in flight on engine `cur` while the next one is being read and uploaded into the other engine.
    uint32_t filled[2] = {0, 0};
    bool in_flight[2] = {false, false};
    int cur = 0;
    auto drain = [&](int i) {
        if (!in_flight[i]) return;
        chk(tm_engine_sync(eng_[i]), "tm_engine_sync");
        for (uint32_t slot = 0; slot < filled[i]; ++slot) {
            const FrameScores r = scores_of(eng_[i], slot);
            if (on_frame) on_frame(r);
            if (s_psnr && r.psnr) s_psnr->push_back(*r.psnr);
            if (s_ssim && r.ssim) s_ssim->push_back(*r.ssim);
            if (s_msssim && r.msssim) s_msssim->push_back(*r.msssim);
            if (s_ssimu && r.ssimulacra2) s_ssimu->push_back(*r.ssimulacra2);
            ++compute_count;
        }
        in_flight[i] = false;
        filled[i] = 0;
    };
    auto submit = [&](int i) {
        if (filled[i] == 0) return;
        chk(tm_engine_compute_async(eng_[i], filled[i]), "tm_engine_compute_async");
        in_flight[i] = true;
    };

    // the two sources are read concurrently: the reference stream on a helper thread, the distorted one here
    HwFrame fref, fdis;
    struct Fetch {
        std::mutex m;
        std::condition_variable cv;
        bool want = false, done = false, quit = false, ok = false, keep = true;
        std::exception_ptr err;
    } fx;
    std::thread helper([&] {
        std::unique_lock<std::mutex> lk(fx.m);
        for (;;) {
            fx.cv.wait(lk, [&] { return fx.want || fx.quit; });
            if (fx.quit) return;
            fx.want = false;
            const bool keep = fx.keep;
            lk.unlock();
            bool ok = false;
            std::exception_ptr err;
            try { ok = keep ? frames_ref.next_frame(fref) : frames_ref.skip_one(); } catch (...) { err = std::current_exception(); }
            lk.lock();
            fx.ok = ok; fx.err = err; fx.done = true;
            fx.cv.notify_all();
        }
    });
    struct Joiner {
        Fetch &f; std::thread &t;
        ~Joiner() { { std::lock_guard<std::mutex> g(f.m); f.quit = true; } f.cv.notify_all(); if (t.joinable()) t.join(); }
    } joiner{fx, helper};
    // keep = false: the pair is consumed but not handed out (dropped by `every`): no upload preparation, and the sources' rings of
    // page-locked surfaces do not advance -- a surface is only reused after `lookahead` KEPT frames, which is what the engines'
    // asynchronous DMA relies on
    auto next_pair = [&](bool keep) {
        { std::lock_guard<std::mutex> g(fx.m); fx.want = true; fx.done = false; fx.keep = keep; }
        fx.cv.notify_all();
        bool ok_dis = false;
        std::exception_ptr err_dis;
        try { ok_dis = keep ? frames_dis.next_frame(fdis) : frames_dis.skip_one(); } catch (...) { err_dis = std::current_exception(); }
        std::unique_lock<std::mutex> lk(fx.m);
        fx.cv.wait(lk, [&] { return fx.done; });
        if (fx.err) std::rethrow_exception(fx.err);
        if (err_dis) std::rethrow_exception(err_dis);
        return fx.ok && ok_dis;
    };
    for (;;) {
        const bool dropped = opts.every > 1 && decode_count != 0 && decode_count % opts.every != 0; // lib.rs:391-394
        auto t0 = tick();
        if (!dropped && kept > UPLOADS_IN_FLIGHT) { // the call below may overwrite the surfaces of pair kept - UPLOADS_IN_FLIGHT - 1
            const Fence &f = fences[(kept - UPLOADS_IN_FLIGHT - 1) % fences.size()]; // (it has one: UPLOADS_IN_FLIGHT >= FENCE_EVERY)
            const int r = tm_engine_upload_done(f.e, f.token, 1);
            if (r < 0) chk(-r, "tm_engine_upload_done");
        }
        timing_.wait_upload += since(t0);
        t0 = tick();
        const bool more = next_pair(!dropped);
        timing_.wait_frames += since(t0);
        if (!more) break;
        if (dropped) {
            ++decode_count;
            continue;
        }
        if (opts.frames > 0 && decode_count >= opts.frames) break; // lib.rs:396-398
        ++decode_count;
        t0 = tick();
        set_frame(eng_[cur], filled[cur], TM_SIDE_REF, fref, cref);
        set_frame(eng_[cur], filled[cur], TM_SIDE_DIS, fdis, cdis);
        ++kept;
        if (kept - first_unfenced >= FENCE_EVERY || filled[cur] + 1 == batch_) { // (a fence covers one engine's uploads: never across a batch)
            Fence f;
            f.e = eng_[cur];
            chk(tm_engine_upload_fence(f.e, &f.token), "tm_engine_upload_fence");
            for (; first_unfenced < kept; ++first_unfenced) fences[first_unfenced % fences.size()] = f;
        }
        timing_.set_frames += since(t0);
        if (++filled[cur] == batch_) {
            t0 = tick();
            submit(cur);
            timing_.submit += since(t0);
            t0 = tick();
            if (eng_[1]) {
                cur ^= 1;
                drain(cur); // the batch submitted before this one: done (or nearly) while we were reading
            } else {
                drain(cur);
            }
            timing_.drain += since(t0);
        }
    }
    submit(cur);
    if (eng_[1]) drain(cur ^ 1);
    drain(cur);
    if (decode_count_out) *decode_count_out = decode_count;

    MetricsResults res;
    res.frame_count = compute_count;
    if (compute_count == 0 && (s_psnr || s_ssim || s_msssim || s_ssimu))
        throw NoFramesSelected();
    if (s_psnr) res.psnr = MetricAggregate::from(std::move(*s_psnr));
    if (s_ssim) res.ssim = MetricAggregate::from(std::move(*s_ssim));
    if (s_msssim) res.msssim = MetricAggregate::from(std::move(*s_msssim));
    if (s_ssimu) res.ssimulacra2 = MetricAggregate::from(std::move(*s_ssimu));
    return res;
}

} // namespace tm_host
