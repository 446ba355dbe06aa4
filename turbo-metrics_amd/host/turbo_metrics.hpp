// turbo_metrics.hpp -- C++ host side of the frame-pair path, above the C ABI (include/turbo_metrics_hip.h).
//
// The reference's host code is Rust; this image has no Rust toolchain, so the host side is C++17 with the
// reference's own names, argument meaning and error behaviour (paths relative to /root/reference/crates):
//   Metrics, Options                       turbo-metrics/src/lib.rs:27-54
//   MetricAggregate, MetricsResults,
//   MetricsStats, FrameScores              turbo-metrics/src/lib.rs:56-123
//   HwFrame, FormatIdentifier, FrameSource turbo-metrics/src/lib.rs:125-186
//   TurboMetrics::{new, metrics,
//     compute_one, compute_all}            turbo-metrics/src/lib.rs:188-433
//   init_cuda                              turbo-metrics/src/lib.rs:438-456   (here init_hip)
//   ColorCharacteristics & fallback, ColorRange, get_color_matrix, get_transfer
//                                          turbo-metrics/src/color.rs:10-94, codec-bitstream/src/lib.rs:98-248
//   Stats                                  quick-stats/src/lib.rs:4-97
// Everything numeric happens in libturbometrics_hip.so; nothing here computes a metric.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/turbo_metrics_hip.h"
#include "quick_stats.hpp"

namespace tm_host {

struct Metrics {
    bool psnr = false, ssim = false, msssim = false, ssimulacra2 = false;
    uint32_t mask() const
    {
        return (psnr ? (uint32_t)TM_METRIC_PSNR : 0u) | (ssim ? (uint32_t)TM_METRIC_SSIM : 0u) |
               (msssim ? (uint32_t)TM_METRIC_MSSSIM : 0u) | (ssimulacra2 ? (uint32_t)TM_METRIC_SSIMULACRA2 : 0u);
    }
};

struct Options {
    uint32_t every = 0, skip = 0, skip_ref = 0, skip_dis = 0, frames = 0;
    // Not a reference field -- frame-pair sharding across devices: this call is one shard of a longer stream and
    // starts at decode index `decode_start` (that many further pairs are skipped first, and the `every` / `frames` arithmetic
    // of lib.rs:391-398 continues from there, so that the shards together select exactly the frames one call would).
    uint32_t decode_start = 0;
};

struct MetricAggregate {
    std::vector<double> scores;
    Stats stats;
    static MetricAggregate from(std::vector<double> v)
    {
        MetricAggregate a;
        a.stats = Stats::compute(v);
        a.scores = std::move(v);
        return a;
    }
};

struct MetricsResults {
    size_t frame_count = 0;
    std::optional<MetricAggregate> psnr, ssim, msssim, ssimulacra2;
};

struct MetricsStats {
    size_t frame_count = 0;
    std::optional<Stats> psnr, ssim, msssim, ssimulacra2;
    static MetricsStats from(const MetricsResults &r)
    {
        MetricsStats s;
        s.frame_count = r.frame_count;
        if (r.psnr) s.psnr = r.psnr->stats;
        if (r.ssim) s.ssim = r.ssim->stats;
        if (r.msssim) s.msssim = r.msssim->stats;
        if (r.ssimulacra2) s.ssimulacra2 = r.ssimulacra2->stats;
        return s;
    }
};

struct FrameScores {
    std::optional<double> psnr, ssim, msssim, ssimulacra2;
};

// ---- colour metadata (H.273 code points the reference understands, codec-bitstream/src/lib.rs:98-248) ----------
enum class ColourPrimaries { Invalid, Unspecified, Unsupported, BT709, BT601_525, BT601_625 };
enum class MatrixCoefficients { Invalid, Unspecified, Unsupported, BT709, BT601_525, BT601_625 };
enum class TransferCharacteristic { Invalid, Unspecified, Unsupported, BT709 };
enum class ColorRange { Limited, Full };

struct ColorCharacteristics {
    ColourPrimaries cp = ColourPrimaries::Unspecified;
    MatrixCoefficients mc = MatrixCoefficients::Unspecified;
    TransferCharacteristic tc = TransferCharacteristic::Unspecified;
    // H.264 table E-3/E-4/E-5 code points as the reference maps them (codec-bitstream/src/h264.rs:104-166, lib.rs:98-248):
    // 1 = BT.709, 2 = unspecified, 5 = BT601_625, 6 = BT601_525 (transfer: 1 and 6 -> BT709), 0 / 3 invalid, rest unsupported
    static ColorCharacteristics from_codes(int cp, int mc, int tc);
    // `.or(fallback)` (codec-bitstream/src/lib.rs:68-95): an Unspecified or Invalid field is taken from `other`
    ColorCharacteristics or_(const ColorCharacteristics &other) const;
};
// turbo-metrics/src/color.rs:51-78: unspecified metadata falls back by frame height
ColorCharacteristics color_characteristics_fallback(uint32_t height);
// color.rs:80-94; combinations the reference leaves as todo!() throw std::runtime_error("not implemented: ...")
int get_color_matrix(const ColorCharacteristics &c);
int get_transfer(const ColorCharacteristics &c);
const char *to_string(ColourPrimaries v);
const char *to_string(MatrixCoefficients v);
const char *to_string(TransferCharacteristic v);
const char *to_string(ColorRange v);

// ---- frames ----------------------------------------------------------------------------------------------------
// the aggregate of per-frame scores in stream order (what compute_all returns; also how the shards of several devices are merged)
MetricsResults aggregate_scores(const std::vector<FrameScores> &frames, const Metrics &metrics);

// One decoded frame handed to the engine (== HwFrame, lib.rs:125-130).  NvDecNV12 / NvDecP016 carry the surface
// contract of an NVDEC mapping (cudarse-video/src/dec.rs:299-403): luma rows at `pitch`, interleaved CbCr at `uv`.
struct HwFrame {
    // Planar420: planar 4:2:0 as files deliver it (not a reference kind: its decoder only yields NV12 / P016) -- data = Y, u, v =
    // the chroma planes at pitch_uv, `bits` = 8, or 9..16 for little-endian u16 samples with the value in the low bits
    // Planar420P10: the same 10-bit planes packed three samples to a 32-bit word (tm_engine_set_frame_i420p10: the upload form; pitches in
    // bytes of the packed rows)
    enum Kind { NvDecNV12, NvDecP016, Npp8, Npp16, Npp32, Planar420, Planar420P10 } kind = Npp8;
    const void *data = nullptr; // luma plane or packed RGB
    const void *uv = nullptr;   // CbCr plane (NvDec kinds)
    const void *u = nullptr, *v = nullptr; // Cb, Cr planes (Planar420)
    size_t pitch = 0;           // bytes
    size_t pitch_uv = 0;        // bytes, chroma rows of Planar420
    int bits = 8;               // Planar420
    bool device = false;        // the pointers are device memory (zero copy) rather than host memory
    bool pinned = false;        // host memory from tm_host_alloc that stays untouched until the engine has synced: async DMA
};

struct FormatIdentifier {
    std::optional<std::string> container;
    std::string codec, decoder;
    std::string str() const { return (container ? *container + "/" : std::string()) + codec + "/" + decoder; }
};

// == trait FrameSource (lib.rs:148-156).  next_frame returns false at end of stream; the frame's memory stays valid
// until the next call on the same source.  Errors are exceptions (the reference returns Box<dyn Error>).
class FrameSource {
public:
    virtual ~FrameSource() = default;
    virtual FormatIdentifier format_id() const = 0;
    virtual uint32_t width() const = 0;
    virtual uint32_t height() const = 0;
    virtual std::pair<ColorCharacteristics, ColorRange> color_characteristics() const = 0;
    virtual size_t frame_count() const = 0; // 0 when unknown
    virtual void skip_frames(uint32_t n) = 0;
    virtual bool next_frame(HwFrame &out) = 0;
    // consume one frame that nobody will look at (`--every N` drops N-1 of N decoded frames, lib.rs:391-394): sources that
    // prepare a frame for upload override this to skip that work and to leave their ring of page-locked surfaces alone
    virtual bool skip_one() { HwFrame f; return next_frame(f); }
    // can a second source opened on the same path jump to an arbitrary frame cheaply?  (regular planar files: yes; images, pipes and
    // decoded video: no) -- what `--devices N` needs to give every device its own shard of the stream
    virtual bool shardable() const { return false; }
    // How many further next_frame calls a returned frame must survive (the engine reads a pinned frame asynchronously until
    // its batch has synced).  Sources that hand out pinned memory size their ring from this; call before the first frame.
    virtual void set_lookahead(size_t frames) {}
    // May the source read pictures AHEAD of the next_frame calls (regular planar files: a pool of readers fills a few ring slots
    // beyond the one being asked for)?  compute_all switches it off when `--every` drops frames: dropped pictures are then
    // consumed without being read at all.  Call before the first frame.
    virtual void set_readahead(bool on) { (void)on; }
    // Allocate what the source needs to hand out frames -- for planar streams the ring of page-locked surfaces, the counterpart of the
    // surface pool the reference's decoder allocates when it is CREATED (cudarse-video/src/dec_simple.rs), i.e. before the CLI's clock
    // starts (turbo-metrics-cli/src/main.rs:252).  Call after set_lookahead / set_readahead; next_frame does it itself otherwise.
    virtual void prepare() {}
};

// CPUs this process may really use: the smallest of the hardware threads, the affinity mask and the cgroup CPU quota (a container
// that sees 256 CPUs may be limited to 16 CPUs' worth of time -- the GPU boxes of rounds 1-4 are: cpu.max = "1600000 100000" --
// and more busy threads than that only buy throttling: the whole group sleeps out the rest of every 100-ms period).
unsigned effective_cpus();
// how many frame sources this process reads at the same time (default 2: reference and distorted; 2 per device with `--devices N`):
// the sources share the usable CPUs between their reader threads
void set_concurrent_streams(unsigned n);

// compute_all selected no frame pair at all (the reference panics in Stats::compute: index out of bounds)
class NoFramesSelected : public std::out_of_range {
public:
    NoFramesSelected() : std::out_of_range("no frame pair was processed (the reference panics in Stats::compute: index out of bounds)") {}
};

class TmError : public std::runtime_error {
public:
    int code;
    TmError(int code, const std::string &where);
};

// Bind the process to `device`; throws when there is no usable gfx950 GPU (there is no CPU path).
void init_hip(int device = 0);

class TurboMetrics {
public:
    // `batch` frame-pair slots per launch (1 == the reference's one pair at a time); `pipeline`: keep a second engine so
    // that reading / uploading the next batch overlaps the current one's kernels (compute_all only)
    TurboMetrics(uint32_t width, uint32_t height, const Metrics &metrics, uint32_t batch = 1, bool pipeline = false);
    ~TurboMetrics();
    TurboMetrics(const TurboMetrics &) = delete;
    TurboMetrics &operator=(const TurboMetrics &) = delete;

    Metrics metrics() const { return metrics_; }
    uint32_t width() const { return w_; }
    uint32_t height() const { return h_; }
    uint32_t batch() const { return batch_; }
    // where the calling thread spent the last compute_all (seconds): waiting for an upload slot, waiting for the two sources, handing
    // frames to the engine (the copies' submission), submitting batches, waiting for results + the per-frame callback
    struct LoopTiming { double wait_upload = 0, wait_frames = 0, set_frames = 0, submit = 0, drain = 0; };
    const LoopTiming &loop_timing() const { return timing_; }
    size_t mem_usage() const;
    // also compute the 56 SSIMULACRA2 sums whose weight is 0.0 (see tm_engine_set_full_sums); the scores do not change
    void set_full_sums(bool on);
    // tuning / measurement (tm_engine_debug_set_param): the results never depend on it
    void debug_set_param(int param, long long value);
    // measurement: how many pairs' uploads may be in flight behind the one being read (4) and how many pairs share a fence (1)
    static void set_upload_tuning(size_t in_flight, size_t fence_every);

    using ColorInfo = std::pair<ColorCharacteristics, ColorRange>;
    // == compute_one (lib.rs:268-360): convert both frames, compute every selected metric, block, return the scores
    FrameScores compute_one(const HwFrame &fref, const ColorInfo &cref, const HwFrame &fdis, const ColorInfo &cdis);

    // compute_one WITHOUT its blocking stream sync (lib.rs:352): hands the pair over, launches, and returns a ticket at once;
    // collect(ticket) blocks until THAT pair's scores are there.  Two launches may be in flight (two engines taking turns -- the
    // second one is created at the first call, the cost of one more engine in device memory): a caller that collects pair k after
    // submitting pair k+1 keeps the device busy while it fetches / decodes the next frames, which is worth 1.6 x at one 1080p pair
    // per call (3.0 k -> 5.0 k pairs/s).  A third submission first finishes the oldest pair and keeps its scores until collected.
    // set_deferred_depth(d), 2 <= d <= 8: d launches in flight on d engines (one pair leaves most of the chip idle: three / four
    // in flight reach 6.8 k / 8.3 k pairs/s with frames in HBM) for a caller that collects pair k after submitting pair k + d - 1.
    // Frames: host memory is read before the call returns unless `pinned` (then until collect(ticket)); device memory until collect.
    // Needs an object created with batch = 1 (throws TmError(TM_ERR_INVALID_ARG) otherwise, like the Python mirror): the second engine has one slot.
    // A setting changed while pairs are in flight (set_full_sums, debug_set_param) first finishes them and keeps their scores for collect.
    // Scores are bit-identical with compute_one's.  Tickets are collected at most once, in any order; compute_one and compute_all
    // may be called in between (they first finish what is in flight and keep those scores for collect).
    uint64_t compute_one_deferred(const HwFrame &fref, const ColorInfo &cref, const HwFrame &fdis, const ColorInfo &cdis);
    FrameScores collect(uint64_t ticket);
    // pairs in flight at most (default 2); pairs in flight are finished first and keep their scores for collect, engines beyond the new
    // depth are freed.  create_now: the engines of the turn are created by this call instead of when their turn first comes (an engine's
    // creation takes 1-20 ms, most beside running launches).  Throws TmError(TM_ERR_INVALID_ARG) outside 2 ... MAX_DEFERRED_DEPTH.
    static constexpr uint32_t MAX_DEFERRED_DEPTH = 8;
    void set_deferred_depth(uint32_t depth, bool create_now = false);
    uint32_t deferred_depth() const { return def_depth_; }

    // == compute_all (lib.rs:362-433) with the frame selection of Options; `on_frame` (optional) sees every FrameScores
    // in stream order (the CLI's output_single_score).  Returns the number of frames decoded (for the CLI's log line).
    MetricsResults compute_all(FrameSource &frames_ref, FrameSource &frames_dis, const Options &opts,
                               const std::function<void(const FrameScores &)> &on_frame = nullptr, uint32_t *decode_count = nullptr);
    // what compute_all asks of its sources (how long a frame must stay valid, whether they may read ahead) + FrameSource::prepare():
    // a caller that times compute_all like the reference's CLI (clock started after the decoders exist) calls this first
    // min_lookahead: a caller that keeps frames longer than compute_all does (compute_one_deferred with d pairs in flight: d - 1 further calls)
    static void prepare_sources(FrameSource &frames_ref, FrameSource &frames_dis, const Options &opts, size_t min_lookahead = 0);

private:
    void set_frame(tm_engine *e, uint32_t slot, int side, const HwFrame &f, const ColorInfo &c);
    FrameScores scores_of(tm_engine *e, uint32_t slot);
    void retire_deferred(); // finish the pairs that are in flight for compute_one_deferred and keep their scores for collect()
    void create_deferred_engine(size_t i);
    uint32_t w_, h_, batch_;
    Metrics metrics_;
    std::vector<tm_engine *> eng_{nullptr, nullptr}; // [0], [1]: compute_all's two; compute_one_deferred takes turns on [0 .. depth)
    LoopTiming timing_;
    // compute_one_deferred: the ticket in flight on each engine (0 = none), finished-but-uncollected scores, next ticket
    uint32_t def_depth_ = 2;
    std::vector<uint64_t> def_pending_{0, 0};
    std::vector<std::pair<uint64_t, FrameScores>> def_done_;
    uint64_t def_next_ = 1;
    bool full_sums_ = false;                                  // settings replayed on an engine that is created later
    std::vector<std::pair<int, long long>> debug_params_;
};

} // namespace tm_host
