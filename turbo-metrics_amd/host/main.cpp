// main.cpp -- `turbo-metrics`: the command line of the reference (crates/turbo-metrics-cli/src/main.rs:31-356) over the
// MI355X engine.  Same positional arguments, same flags (-m/--metrics, --every, --skip, --skip-ref, --skip-dis, --frames,
// --output), same stdout formats (output.cpp), status on stderr, ExitCode::FAILURE on the same conditions.
// Additions (no counterpart in the reference, all optional): --batch, --device, --devices, --ranks, --no-pipeline, --full-sums, and the
// description of headerless YUV input (--width, --height, --bits, --color-primaries, --matrix-coefficients,
// --transfer-characteristics, --full-range).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "frame_sources.hpp"
#include "output.hpp"
#include "ranks.hpp"
#include "turbo_metrics.hpp"

using namespace tm_host;

namespace {

enum Level { L_ERROR = 0, L_WARN, L_INFO, L_DEBUG, L_TRACE };
Level g_level = L_INFO;

void log_line(Level lv, const char *target, const std::string &msg)
{
    if (lv > g_level) return;
    static const char *names[] = {"ERROR", " WARN", " INFO", "DEBUG", "TRACE"};
    std::cerr << names[lv] << " " << target << ": " << msg << "\n"; // tracing fmt::layer().compact().without_time()
}

const char *kTarget = "turbo_metrics_cli";

void usage(std::ostream &os)
{
    os << "Turbo metrics compares two images or videos using quality metrics\n\n"
          "Usage: turbo-metrics [OPTIONS] <REFERENCE> <DISTORTED>\n\n"
          "Arguments:\n"
          "  <REFERENCE>  Reference media (PNG / PPM / PFM image, Y4M or raw planar YUV, IVF / Matroska video through TM_DECODER). Use `-` to read from stdin\n"
          "  <DISTORTED>  Distorted media. Use `-` to read from stdin\n\n"
          "Options:\n"
          "  -m, --metrics <METRICS>    Select the metrics to compute [possible values: psnr, ssim, msssim, ssimulacra2]\n"
          "      --every <EVERY>        Only compute metrics every few frames [default: 0]\n"
          "      --skip <SKIP>          Index of the first frame to start computing at [default: 0]\n"
          "      --skip-ref <SKIP_REF>  Index of the first reference frame, additive with `skip` [default: 0]\n"
          "      --skip-dis <SKIP_DIS>  Index of the first distorted frame, additive with `skip` [default: 0]\n"
          "      --frames <FRAMES>      Amount of frames to compute [default: 0]\n"
          "      --output <OUTPUT>      stdout format [possible values: default, json, json-lines, csv]\n"
          "      --batch <N>            frame pairs per GPU launch [default: by picture size, 16 at 1080p, 8 at 4K]\n"
          "      --device <N>           GPU ordinal [default: 0]\n"
          "      --devices <N>          shard the frame pairs of two regular files over N GPUs (0 = all visible) [default: 1]\n"
          "      --ranks <N>            the same shards as N processes, one per GPU, scores gathered by ONE RCCL reduce to rank 0\n"
          "                             (TM_RANK_TRANSPORT=pipe: over pipes instead; TM_RANK_TIMEOUT_S: give up after that many seconds)\n"
          "      --no-pipeline          do not overlap reading/upload of the next batch with the current one\n"
          "      --loop <MODE>          batched (default: compute_all), reference (the reference's loop: one blocking compute_one per pair),\n"
          "                             deferred (that loop with compute_one_deferred + collect: --in-flight pairs in flight)\n"
          "      --in-flight <N>        pairs in flight with --loop deferred, 2 ... 8: pair k is collected after pair k + N - 1 went in [default: 2]\n"
          "      --full-sums            compute all 108 SSIMULACRA2 sums, also the zero-weighted ones\n"
          "      --width <W> --height <H> [--bits 8|10|12|16]   headerless planar 4:2:0 input\n"
          "      --color-primaries <N> --matrix-coefficients <N> --transfer-characteristics <N>   H.273 codes (1, 5, 6; 2 = by height)\n"
          "      --full-range           the YUV input is full range (unsupported by the reference and here)\n"
          "  -h, --help                 Print help\n"
          "  -V, --version              Print version\n";
}

bool parse_u32(const std::string &s, uint32_t &out)
{
    if (s.empty()) return false;
    char *end = nullptr;
    const unsigned long v = strtoul(s.c_str(), &end, 10);
    if (*end || v > 0xFFFFFFFFul || s[0] == '-') return false;
    out = (uint32_t)v;
    return true;
}

std::string format_duration(std::chrono::milliseconds d) // main.rs:357-385
{
    long long secs = d.count() / 1000;
    const long long millis = d.count() % 1000, minutes = secs / 60;
    secs %= 60;
    std::string s;
    if (minutes > 0) { s += std::to_string(minutes) + " m"; if (secs > 0) s += " "; }
    if (secs > 0) { s += std::to_string(secs) + " s"; if (millis > 0) s += " "; }
    if (millis > 0) s += std::to_string(millis) + " ms";
    return s;
}

void log_source(const char *target, const FrameSource &src)
{
    const auto cc = src.color_characteristics();
    log_line(L_INFO, target, "codec=" + src.format_id().str() + " width=" + std::to_string(src.width()) + " height=" + std::to_string(src.height()) +
                                 " cp=" + to_string(cc.first.cp) + " mc=" + to_string(cc.first.mc) + " tc=" + to_string(cc.first.tc) +
                                 " cr=" + to_string(cc.second) + " frame_count=" + std::to_string(src.frame_count()));
}

} // namespace

// --batch not given: at most 72 Mpx per launch, 2 .. 16 pairs (16 at 1080p and below, 8 at 4K; round 6: with 10-bit pictures packed on the
// link the 4K loop is no longer bound by PCIe alone -- 4 pairs per launch 1 246 pairs/s, 16: 1 333, profiles/r06b_bench.json).  Host-fed streams are
// bound by the file reads and PCIe long before the engine (1080p: ~7.5 k pairs/s over PCIe against 12.7 k pairs/s of the engine at 16
// pairs per launch; 4K: 1.1 k against 3.0 k at 4), and since round 4 the page-locked frame rings no longer grow with the batch
// (upload fences), so the batch only has to be large enough for the kernels and small enough for the engines' memory
// (two engines x batch x 0.24 GB at 1080p, x 0.95 GB at 4K).
static uint32_t auto_batch(uint32_t w, uint32_t h)
{
    const double mpx = (double)w * (double)h / 1e6;
    uint32_t b = 16;
    while (b > 2 && (double)b * mpx > 72.0) b /= 2;
    return b;
}

int main(int argc, char **argv)
{
    if (const char *env = getenv("RUST_LOG")) {
        const std::string v = env;
        if (v == "error") g_level = L_ERROR; else if (v == "warn") g_level = L_WARN; else if (v == "info") g_level = L_INFO;
        else if (v == "debug") g_level = L_DEBUG; else if (v == "trace") g_level = L_TRACE;
    }
    std::vector<std::string> pos;
    Metrics metrics;
    Options opts;
    Output output = Output::Default;
    SourceHints hints;
    uint32_t batch = 0 /* 0: chosen from the picture size */, device = 0, devices = 1, ranks = 0 /* 0: not asked for */, in_flight_pairs = 2;
    bool pipeline = true, full_sums = false, in_flight_given = false;
    enum class Loop { Batched, Reference, Deferred } loop = Loop::Batched;
    std::vector<std::pair<int, long long>> tune;

    auto bad = [&](const std::string &m) {
        std::cerr << "error: " << m << "\n\nFor more information, try '--help'.\n";
        return 2; // clap's usage-error exit code
    };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], val;
        bool has_val = false;
        if (a.rfind("--", 0) == 0) {
            const size_t eq = a.find('=');
            if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_val = true; }
        }
        auto value = [&](std::string &dst) {
            if (has_val) { dst = val; return true; }
            if (i + 1 >= argc) return false;
            dst = argv[++i];
            return true;
        };
        auto u32 = [&](uint32_t &dst) {
            std::string s;
            return value(s) && parse_u32(s, dst);
        };
        if (a == "-h" || a == "--help") { usage(std::cout); return 0; }
        if (a == "-V" || a == "--version") { std::cout << "turbo-metrics " << tm_version() << "\n"; return 0; }
        if (a == "-m" || a == "--metrics") {
            std::string s;
            if (!value(s)) return bad("a value is required for '--metrics <METRICS>' but none was supplied");
            if (s == "psnr") metrics.psnr = true; else if (s == "ssim") metrics.ssim = true; else if (s == "msssim") metrics.msssim = true;
            else if (s == "ssimulacra2") metrics.ssimulacra2 = true;
            else return bad("invalid value '" + s + "' for '--metrics <METRICS>'\n  [possible values: psnr, ssim, msssim, ssimulacra2]");
        } else if (a.rfind("-m", 0) == 0 && a.size() > 2 && a[1] == 'm') { // -mpsnr
            const std::string s = a.substr(2);
            if (s == "psnr") metrics.psnr = true; else if (s == "ssim") metrics.ssim = true; else if (s == "msssim") metrics.msssim = true;
            else if (s == "ssimulacra2") metrics.ssimulacra2 = true;
            else return bad("invalid value '" + s + "' for '--metrics <METRICS>'");
        } else if (a == "--every") { if (!u32(opts.every)) return bad("invalid value for '--every <EVERY>'"); }
        else if (a == "--skip") { if (!u32(opts.skip)) return bad("invalid value for '--skip <SKIP>'"); }
        else if (a == "--skip-ref") { if (!u32(opts.skip_ref)) return bad("invalid value for '--skip-ref <SKIP_REF>'"); }
        else if (a == "--skip-dis") { if (!u32(opts.skip_dis)) return bad("invalid value for '--skip-dis <SKIP_DIS>'"); }
        else if (a == "--frames") { if (!u32(opts.frames)) return bad("invalid value for '--frames <FRAMES>'"); }
        else if (a == "--output") {
            std::string s;
            if (!value(s) || !parse_output(s, output)) return bad("invalid value '" + s + "' for '--output <OUTPUT>'\n  [possible values: default, json, json-lines, csv]");
        } else if (a == "--batch") { if (!u32(batch) || batch == 0) return bad("invalid value for '--batch <N>'"); }
        else if (a == "--device") { if (!u32(device)) return bad("invalid value for '--device <N>'"); }
        else if (a == "--devices") { if (!u32(devices)) return bad("invalid value for '--devices <N>'"); }
        else if (a == "--ranks") { if (!u32(ranks) || ranks == 0 || ranks > 64) return bad("invalid value for '--ranks <N>'"); }
        else if (a == "--no-pipeline") pipeline = false;
        else if (a == "--in-flight") { if (!u32(in_flight_pairs) || in_flight_pairs < 2 || in_flight_pairs > TurboMetrics::MAX_DEFERRED_DEPTH) return bad("invalid value for '--in-flight <N>'\n  [2 ... 8]"); in_flight_given = true; }
        else if (a == "--loop") {
            std::string s;
            if (!value(s)) return bad("a value is required for '--loop <MODE>'");
            if (s == "batched") loop = Loop::Batched; else if (s == "reference") loop = Loop::Reference; else if (s == "deferred") loop = Loop::Deferred;
            else return bad("invalid value '" + s + "' for '--loop <MODE>'\n  [possible values: batched, reference, deferred]");
        }
        else if (a == "--full-sums") full_sums = true;
        else if (a == "--tune") { // --tune <param>=<value>: tm_engine_debug_set_param (measurements; not in the usage text)
            std::string kv; if (!value(kv) || kv.find('=') == std::string::npos) return bad("invalid value for '--tune <param>=<value>'");
            tune.emplace_back(atoi(kv.c_str()), atoll(kv.c_str() + kv.find('=') + 1));
        }
        else if (a == "--width") { if (!u32(hints.width)) return bad("invalid value for '--width <W>'"); }
        else if (a == "--height") { if (!u32(hints.height)) return bad("invalid value for '--height <H>'"); }
        else if (a == "--bits") { uint32_t b; if (!u32(b) || (b != 8 && b != 10 && b != 12 && b != 16)) return bad("invalid value for '--bits'"); hints.bits = (int)b; }
        else if (a == "--color-primaries") { uint32_t v; if (!u32(v)) return bad("invalid value for '--color-primaries'"); hints.cp = (int)v; }
        else if (a == "--matrix-coefficients") { uint32_t v; if (!u32(v)) return bad("invalid value for '--matrix-coefficients'"); hints.mc = (int)v; }
        else if (a == "--transfer-characteristics") { uint32_t v; if (!u32(v)) return bad("invalid value for '--transfer-characteristics'"); hints.tc = (int)v; }
        else if (a == "--full-range") hints.full_range = true;
        else if (a == "--raw") hints.force_raw = true;
        else if (a.size() > 1 && a[0] == '-' && a != "-") return bad("unexpected argument '" + a + "' found");
        else pos.push_back(a);
    }
    if (pos.size() != 2) return bad("the following required arguments were not provided:\n  <REFERENCE>\n  <DISTORTED>");

    const bool ref_is_stdin = pos[0] == "-", dis_is_stdin = pos[1] == "-";
    if (ref_is_stdin && dis_is_stdin) {
        log_line(L_ERROR, kTarget, "Can't read both reference and distorted from stdin");
        return EXIT_FAILURE;
    }

    // ---- --ranks N: one process per GPU and ONE RCCL reduce of the per-frame scores (ranks.hpp).  This process is either the launcher
    // (it makes no GPU call: it starts the N ranks and waits for them) or, with TM_RANK in the environment, one of the ranks.
    RankEnv renv;
    bool is_rank = false;
    try { is_rank = rank_env(renv); }
    catch (const std::exception &e) { log_line(L_ERROR, kTarget, std::string("Could not join the ranks : ") + e.what()); return EXIT_FAILURE; }
    if (ranks > 0 && !is_rank) {
        if (ref_is_stdin || dis_is_stdin) { log_line(L_ERROR, kTarget, "--ranks needs two regular planar-YUV files of known length, not stdin"); return EXIT_FAILURE; }
        if (loop != Loop::Batched) { log_line(L_ERROR, kTarget, "--loop reference / deferred run on one device: not with --ranks"); return EXIT_FAILURE; }
        if (devices != 1) { log_line(L_ERROR, kTarget, "--ranks (one process per GPU) and --devices (one process, N GPUs) exclude each other"); return EXIT_FAILURE; }
        if (metrics.mask() == 0) { log_line(L_ERROR, kTarget, "Could not initialize engine : no metric selected (-m psnr|ssim|msssim|ssimulacra2)"); return EXIT_FAILURE; }
        const char *to = getenv("TM_RANK_TIMEOUT_S");
        std::cout.flush();
        return launch_ranks(argv, (int)ranks, to ? atof(to) : 0.0);
    }
    if (is_rank && (ranks == 0 || (int)ranks != renv.world)) { log_line(L_ERROR, kTarget, "TM_RANK is set but --ranks does not match TM_WORLD"); return EXIT_FAILURE; }
    if (is_rank && renv.rank != 0 && g_level > L_WARN && !getenv("RUST_LOG")) g_level = L_WARN; // rank 0 tells the story once

    // The reference initialises the device first because its frame sources decode on it (main.rs:138-139).  Here the sources
    // come first: a compressed-video source starts its decoder as a child process, and that must happen before this process
    // has touched the GPU runtime (a source only page-locks memory when its first frame is asked for, after init_hip).
    std::unique_ptr<FrameSource> source_ref, source_dis;
    try {
        source_ref = create_source(pos[0], hints);
    } catch (const std::exception &e) {
        log_line(L_ERROR, kTarget, std::string("Could not read reference : ") + e.what());
        return EXIT_FAILURE;
    }
    try {
        source_dis = create_source(pos[1], hints);
    } catch (const std::exception &e) {
        log_line(L_ERROR, kTarget, std::string("Could not read distorted : ") + e.what());
        return EXIT_FAILURE;
    }
    if (source_ref->width() != source_dis->width() || source_ref->height() != source_dis->height()) {
        // the reference logs this and carries on into undefined territory (main.rs:156-158); here it is fatal
        log_line(L_ERROR, kTarget, "Reference and distorted are not the same size");
        return EXIT_FAILURE;
    }
    if (is_rank) { // rank r works on device (device + r); TM_SHARE_DEVICE=1 (tests on a 1-GPU box) lets the ranks share what is visible
        const int visible = std::max(1, tm_device_count());
        const bool share = getenv("TM_SHARE_DEVICE") && atoi(getenv("TM_SHARE_DEVICE")) != 0;
        if (!share && renv.world > visible) {
            if (renv.rank == 0) log_line(L_ERROR, kTarget, "--ranks " + std::to_string(renv.world) + " but only " + std::to_string(visible) + " GPU(s) visible");
            return EXIT_FAILURE;
        }
        device = (device + (uint32_t)renv.rank) % (uint32_t)visible;
    }
    try {
        init_hip((int)device);
        if (g_level >= L_TRACE) // the engine library's own diagnostics (placement search, hardware queues): a function of the laboratory build
            if (auto fn = (void (*)(int))dlsym(RTLD_DEFAULT, "tm_set_debug_log")) fn(1);
    } catch (const std::exception &e) {
        log_line(L_ERROR, kTarget, std::string("Could not initialize the GPU : ") + e.what());
        return EXIT_FAILURE;
    }

    // One block of decode indices [lo, hi) on the device this thread is bound to: its own sources (opened on the same files), its own
    // engines, the reference's selection loop (lib.rs:385-404) with decode_start = lo.  Shared by --devices (a thread per device) and
    // --ranks (a process per device).
    struct Block { std::vector<FrameScores> scores; uint32_t decoded = 0; };
    // ready (optional): called once the block's engines exist and its sources have page-locked their rings -- where the one-device run starts its
    // clock (main.rs:252: after the decoders and the engine exist)
    // each: sees every FrameScores of the block as it arrives (rank 0 prints its own block while the others compute)
    // computed (optional): called when the block's last score is there, before its engines and rings are given back
    const auto score_block = [&](uint32_t w, uint32_t h, uint32_t lo, uint32_t hi, Block &out, const std::function<void()> &ready = nullptr,
                                 const std::function<void(const FrameScores &)> &each = nullptr, const std::function<void()> &computed = nullptr) {
        if (lo >= hi) { if (ready) ready(); if (computed) computed(); return; }
        auto sr = create_source(pos[0], hints), sd = create_source(pos[1], hints);
        const uint32_t b = std::min(batch, hi - lo);
        TurboMetrics tmx(w, h, metrics, b, pipeline && hi - lo > b);
        if (full_sums) tmx.set_full_sums(true);
        for (auto &t : tune) if (t.first < 100) tmx.debug_set_param(t.first, t.second);
        Options o = opts;
        o.decode_start = lo;
        o.frames = hi; // absolute decode index at which this block stops (lib.rs:396-398)
        if (ready) { TurboMetrics::prepare_sources(*sr, *sd, o); ready(); }
        uint32_t dc = lo;
        try { tmx.compute_all(*sr, *sd, o, [&](const FrameScores &fs) { out.scores.push_back(fs); if (each) each(fs); }, &dc); }
        catch (const NoFramesSelected &) { dc = hi; } // a block in which `every` selects no frame is empty, not an error (any other exception is one)
        out.decoded = dc - lo;
        if (computed) computed();
        if (g_level >= L_DEBUG) {
            const TurboMetrics::LoopTiming &t = tmx.loop_timing();
            char b[256];
            snprintf(b, sizeof b, "block [%u, %u): %.0f ms waiting for an upload slot, %.0f ms for the sources, %.0f ms handing frames to the engine, %.0f ms submitting, %.0f ms waiting for results + output",
                     lo, hi, t.wait_upload * 1e3, t.wait_frames * 1e3, t.set_frames * 1e3, t.submit * 1e3, t.drain * 1e3);
            log_line(L_DEBUG, kTarget, b);
        }
    };
    // what the one-device run prints behind the last per-frame line
    const auto report = [&](const std::vector<FrameScores> &all, uint32_t decoded, uint32_t w, uint32_t h, std::chrono::steady_clock::time_point start, const std::string &where) -> int {
        MetricsResults results;
        try { results = aggregate_scores(all, metrics); }
        catch (const std::exception &e) { std::cout.flush(); log_line(L_ERROR, kTarget, std::string("Computation failed : ") + e.what()); return EXIT_FAILURE; }
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start);
        const long long dms = ms.count() > 0 ? ms.count() : 1;
        const unsigned long long fps = (unsigned long long)results.frame_count * 1000ull / (unsigned long long)dms;
        char perf_s[64];
        snprintf(perf_s, sizeof perf_s, "%.3f", (double)w * (double)h * (double)results.frame_count / (double)dms / 1000.0);
        log_line(L_INFO, kTarget, "Processed: " + std::to_string(results.frame_count) + " (decoded: ~" + std::to_string(decoded + opts.skip) +
                                      ") frame pairs in " + format_duration(ms) + " (" + std::to_string(fps) + " fps) (Mpx/s: " + perf_s + ") on " + where);
        output_results(output, results, std::cout);
        std::cout.flush();
        return EXIT_SUCCESS;
    };

    if (is_rank) {
        const bool root = renv.rank == 0;
        const size_t known = std::min(source_ref->frame_count(), source_dis->frame_count());
        const uint32_t lead = opts.skip + std::max(opts.skip_ref, opts.skip_dis);
        if (known == 0 || known <= lead || !source_ref->shardable() || !source_dis->shardable()) {
            if (root) log_line(L_ERROR, kTarget, "--ranks needs two regular planar-YUV files of known length");
            return EXIT_FAILURE;
        }
        uint32_t total = (uint32_t)(known - lead); // decode indices available
        if (opts.frames > 0) total = std::min(total, opts.frames);
        if (root) { log_source("reference", *source_ref); log_source("distorted", *source_dis); }
        const uint32_t w = source_ref->width(), h = source_ref->height();
        if (batch == 0) batch = auto_batch(w, h);
        set_concurrent_streams(2 * (unsigned)renv.world); // the ranks' reader threads share the usable CPUs of the node
        source_ref.reset(); source_dis.reset();
        if (known < 20000) tm_set_placement_candidates(1);
        try {
            // The transport is made AFTER the block's engines (inside `ready`, which every rank reaches exactly once): the runtime binds a stream to
            // one of its four hardware queues when the stream is created, and a communicator that exists first -- RCCL creates streams of its own --
            // takes the queues the engine's side and upload streams would have had to themselves (profiles/r06y6_queue_map.log; `--ranks 1` over
            // RCCL then ran at 0.83 x the one-device rate, profiles/r06y8_ranks_order.log).
            std::unique_ptr<RankTransport> transport;
            auto start = std::chrono::steady_clock::now();
            uint32_t lo = 0, hi = 0;
            shard_range(total, (uint32_t)renv.rank, (uint32_t)renv.world, lo, hi);
            Block blk;
            // rank 0's clock starts where the one-device run's does: engines created, rings page-locked (the other ranks set up at the same time; the
            // figure then covers rank 0's block, the wait for the slowest rank and the reduce)
            // rank 0 holds the FIRST block: its lines go out as they are computed, like the one-device run's; the other blocks' follow the reduce
            if (root) output_prepare(output, metrics, std::cout);
            // (the clock is stopped while the block's engines and page-locked rings are freed: the one-device run frees its own after its report)
            auto computed_at = start;
            score_block(w, h, lo, hi, blk,
                        [&] {
                            transport = make_rank_transport(renv);
                            if (root) log_line(L_DEBUG, kTarget, std::string("ranks: ") + std::to_string(renv.world) + " over " + transport->name());
                            start = std::chrono::steady_clock::now();
                        },
                        root ? std::function<void(const FrameScores &)>([&](const FrameScores &fs) { output_single_score(output, fs, std::cout); }) : nullptr,
                        [&] { computed_at = std::chrono::steady_clock::now(); });
            start += std::chrono::steady_clock::now() - computed_at;
            ScoreVector sv(metrics, total);
            size_t k = 0;
            for (uint32_t dc = lo; dc < hi && k < blk.scores.size(); ++dc) {
                if (opts.every > 1 && dc != 0 && dc % opts.every != 0) continue; // lib.rs:391-394
                sv.put(dc, blk.scores[k++]);
            }
            if (k != blk.scores.size()) throw std::runtime_error("a block produced more scores than it has selected frames");
            sv.add_decoded(blk.decoded);
            transport->reduce_sum_to_root(sv.v); // the ONE collective of the path
            if (!root) return EXIT_SUCCESS;
            const std::vector<FrameScores> all = sv.frames();
            for (size_t i = blk.scores.size(); i < all.size(); ++i) output_single_score(output, all[i], std::cout); // (rank 0's own are out already)
            return report(all, sv.decoded(), w, h, start, std::to_string(renv.world) + " ranks (" + transport->name() + ")");
        } catch (const RankPeerLost &e) { // not this rank's failure: the launcher reports the rank that went away
            std::cout.flush();
            log_line(L_WARN, kTarget, "rank " + std::to_string(renv.rank) + " gives up : " + e.what());
            return RANK_PEER_LOST;
        } catch (const std::exception &e) {
            std::cout.flush();
            log_line(L_ERROR, kTarget, "Computation failed (rank " + std::to_string(renv.rank) + ") : " + e.what());
            return EXIT_FAILURE;
        }
    }

    // ---- frame-pair sharding over several GPUs (the reference is single-GPU: device 0 hard-coded, lib.rs:442).
    // One host thread per device, each with its own sources (opened on the same files) and its own engines; decode indices
    // [0, L) are cut into contiguous blocks, every thread runs the reference's selection loop on its block, and the per-frame
    // scores are concatenated in block order: the output is byte-identical to the single-device run.  (bench.py measures the
    // one-process-per-GPU arrangement with the RCCL reduce; inside one process the "reduce" is this concatenation.)
    {
        int visible = tm_device_count();
        if (visible < 1) visible = 1;
        // TM_SHARE_DEVICE=1 (tests on a 1-GPU box): the shards share the visible devices round-robin
        const bool share = getenv("TM_SHARE_DEVICE") && atoi(getenv("TM_SHARE_DEVICE")) != 0;
        uint32_t want = devices == 0 ? (uint32_t)visible : devices;
        if (want > 1 && !share && want > (uint32_t)visible) {
            log_line(L_ERROR, kTarget, "--devices " + std::to_string(want) + " but only " + std::to_string(visible) + " GPU(s) visible");
            return EXIT_FAILURE;
        }
        const size_t known = std::min(source_ref->frame_count(), source_dis->frame_count());
        const uint32_t lead = opts.skip + std::max(opts.skip_ref, opts.skip_dis);
        if (want > 1 && (ref_is_stdin || dis_is_stdin || known == 0 || known <= lead || !source_ref->shardable() || !source_dis->shardable())) {
            log_line(L_WARN, kTarget, "--devices needs two regular planar-YUV files of known length: running on one device");
            want = 1;
        }
        if (want > 1 && loop != Loop::Batched) log_line(L_WARN, kTarget, "--loop reference / deferred run on one device: --devices ignored");
        if (in_flight_given && loop != Loop::Deferred) log_line(L_WARN, kTarget, "--in-flight belongs to --loop deferred: ignored");
        if (want > 1 && loop != Loop::Batched) want = 1;
        if (want > 1) {
            if (metrics.mask() == 0) { log_line(L_ERROR, kTarget, "Could not initialize engine : no metric selected (-m psnr|ssim|msssim|ssimulacra2)"); return EXIT_FAILURE; }
            uint32_t total = (uint32_t)(known - lead); // decode indices available
            if (opts.frames > 0) total = std::min(total, opts.frames);
            want = std::min(want, std::max(1u, total));
            const uint32_t per = (total + want - 1) / want;
            log_source("reference", *source_ref);
            log_source("distorted", *source_dis);
            const uint32_t w = source_ref->width(), h = source_ref->height();
            if (batch == 0) batch = auto_batch(w, h);
            set_concurrent_streams(2 * want); // every shard reads its own pair of streams: the reader threads share the usable CPUs
            source_ref.reset(); source_dis.reset(); // every shard opens its own
            if (known < 20000) tm_set_placement_candidates(1); // as below: the search pays off on long streams only
            struct Shard { Block blk; std::string err; };
            std::vector<Shard> shards(want);
            const auto start = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (uint32_t r = 0; r < want; ++r)
                th.emplace_back([&, r] {
                    Shard &sh = shards[r];
                    try {
                        const uint32_t lo = std::min(total, r * per), hi = std::min(total, lo + per);
                        if (lo >= hi) return;
                        init_hip((int)((device + r) % (uint32_t)std::max(1, visible)));
                        score_block(w, h, lo, hi, sh.blk);
                    } catch (const std::exception &e) { sh.err = e.what(); }
                });
            for (auto &t : th) t.join();
            for (const Shard &sh : shards)
                if (!sh.err.empty()) { log_line(L_ERROR, kTarget, "Computation failed : " + sh.err); return EXIT_FAILURE; }
            output_prepare(output, metrics, std::cout);
            std::vector<FrameScores> all;
            uint32_t decoded = 0;
            for (const Shard &sh : shards) {
                for (const FrameScores &fs : sh.blk.scores) { output_single_score(output, fs, std::cout); all.push_back(fs); }
                decoded += sh.blk.decoded;
            }
            return report(all, decoded, w, h, start, std::to_string(want) + " devices");
        }
    }

    std::unique_ptr<TurboMetrics> turbo;
    try {
        if (metrics.mask() == 0) throw std::runtime_error("no metric selected (-m psnr|ssim|msssim|ssimulacra2)");
        // a source that knows its length never needs more slots than it has pairs (a single image pair: one slot, one engine)
        const size_t known = std::min(source_ref->frame_count(), source_dis->frame_count());
        if (batch == 0) batch = auto_batch(source_ref->width(), source_ref->height());
        if (known > 0 && known <= batch) { batch = (uint32_t)known; pipeline = false; }
        if (loop != Loop::Batched) { batch = 1; pipeline = false; } // one pair per call, like the reference
        // the placement search of tm_engine_create (~10 ms per candidate and engine) pays off on long streams only
        if (known > 0 && known < 20000) tm_set_placement_candidates(1);
        turbo = std::make_unique<TurboMetrics>(source_ref->width(), source_ref->height(), metrics, batch, pipeline);
        if (full_sums) turbo->set_full_sums(true);
        for (auto &t : tune) if (t.first < 100) turbo->debug_set_param(t.first, t.second);
        // --loop deferred: every engine of the turn exists before the clock starts (the reference allocates everything up front too,
        // ssimulacra2-cuda/src/lib.rs:21)
        if (loop == Loop::Deferred) turbo->set_deferred_depth(in_flight_pairs, true);
    } catch (const std::exception &e) {
        log_line(L_ERROR, kTarget, std::string("Could not initialize engine : ") + e.what());
        return EXIT_FAILURE;
    }

    log_source("reference", *source_ref);
    log_source("distorted", *source_dis);
    // the sources' frame rings are page-locked here, before the clock starts: the counterpart of the surface pool the reference's decoder
    // allocates when it is created (its clock, main.rs:252, starts after decoders and engine exist too)
    {   // --tune 100=<pairs in flight> / 101=<pairs per fence>: host-side upload tuning (the rest goes to the engines)
        size_t in_flight = 4, fence_every = 1;
        bool given = false;
        for (auto &t : tune) { if (t.first == 100) { in_flight = (size_t)t.second; given = true; } if (t.first == 101) { fence_every = (size_t)t.second; given = true; } }
        if (given) TurboMetrics::set_upload_tuning(in_flight, fence_every); // (default: by picture size, TurboMetrics::prepare_sources)
    }
    try { TurboMetrics::prepare_sources(*source_ref, *source_dis, opts, loop == Loop::Deferred ? in_flight_pairs : 0); }
    catch (const std::exception &e) { log_line(L_ERROR, kTarget, std::string("Could not initialize the sources : ") + e.what()); return EXIT_FAILURE; }
    log_line(L_DEBUG, kTarget, "Initialized, now processing ...");

    const auto start = std::chrono::steady_clock::now();
    output_prepare(output, metrics, std::cout);
    MetricsResults results;
    uint32_t decode_count = 0;
    try {
        if (loop == Loop::Batched) {
            results = turbo->compute_all(*source_ref, *source_dis, opts,
                                         [&](const FrameScores &r) { output_single_score(output, r, std::cout); }, &decode_count);
        } else {
            // The reference's own loop, statement for statement (turbo-metrics-cli/src/main.rs:284-326): one pair per call.  `deferred`
            // is the same loop with the two-line change INTEGRATION.md section 3 shows: submit pair k, then collect pair k - 1.
            const auto cref = source_ref->color_characteristics(), cdis = source_dis->color_characteristics();
            std::vector<FrameScores> all;
            const auto emit = [&](const FrameScores &r) { output_single_score(output, r, std::cout); all.push_back(r); };
            source_ref->skip_frames(opts.skip_ref + opts.skip);
            source_dis->skip_frames(opts.skip_dis + opts.skip);
            HwFrame fref, fdis;
            std::deque<uint64_t> tickets; // --loop deferred: pairs in flight, oldest first
            while (source_ref->next_frame(fref) && source_dis->next_frame(fdis)) {
                if (opts.every > 1 && decode_count != 0 && decode_count % opts.every != 0) { decode_count += 1; continue; }
                if (opts.frames > 0 && decode_count >= opts.frames) break;
                decode_count += 1;
                if (loop == Loop::Reference) {
                    emit(turbo->compute_one(fref, cref, fdis, cdis));
                } else {
                    tickets.push_back(turbo->compute_one_deferred(fref, cref, fdis, cdis));
                    if (tickets.size() >= in_flight_pairs) { emit(turbo->collect(tickets.front())); tickets.pop_front(); }
                }
            }
            for (uint64_t t : tickets) emit(turbo->collect(t));
            results = aggregate_scores(all, metrics);
        }
    } catch (const std::exception &e) {
        std::cout.flush();
        log_line(L_ERROR, kTarget, std::string("Computation failed : ") + e.what());
        return EXIT_FAILURE;
    }
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start);
    const long long dms = ms.count() > 0 ? ms.count() : 1; // the reference divides by as_millis() (main.rs:332)
    const unsigned long long fps = (unsigned long long)results.frame_count * 1000ull / (unsigned long long)dms;
    const double perf = (double)source_ref->width() * (double)source_ref->height() * (double)results.frame_count / (double)dms / 1000.0;
    char perf_s[64];
    snprintf(perf_s, sizeof perf_s, "%.3f", perf);
    log_line(L_INFO, kTarget, "Processed: " + std::to_string(results.frame_count) + " (decoded: ~" + std::to_string(decode_count + opts.skip) +
                                  ") frame pairs in " + format_duration(ms) + " (" + std::to_string(fps) + " fps) (Mpx/s: " + perf_s + ")");
    {
        const TurboMetrics::LoopTiming &t = turbo->loop_timing();
        char b[256];
        snprintf(b, sizeof b, "main thread: %.0f ms waiting for an upload slot, %.0f ms for the sources, %.0f ms handing frames to the engine, %.0f ms submitting, %.0f ms waiting for results + output",
                 t.wait_upload * 1e3, t.wait_frames * 1e3, t.set_frames * 1e3, t.submit * 1e3, t.drain * 1e3);
        log_line(L_DEBUG, kTarget, b);
    }
    output_results(output, results, std::cout);
    std::cout.flush();
    return EXIT_SUCCESS;
}
