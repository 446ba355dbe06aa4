// tests/host/tm_host_test.cpp -- TEST INFRASTRUCTURE: exposes the host-side pieces of turbo-metrics_amd/host that need no
// GPU (number formatting, Stats, image / Y4M decoding, output layer) to the pytest tier through a tiny command line.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../turbo-metrics_amd/host/frame_sources.hpp"
#include "../../turbo-metrics_amd/host/video_input.hpp"
#include "../../turbo-metrics_amd/host/output.hpp"
#include "../../turbo-metrics_amd/host/rust_fmt.hpp"
#include "../../turbo-metrics_amd/host/ranks.hpp"
#include <thread>
#include <unistd.h>

using namespace tm_host;

static double parse_double(const std::string &s)
{
    if (s == "inf") return INFINITY;
    if (s == "-inf") return -INFINITY;
    if (s == "nan") return NAN;
    return strtod(s.c_str(), nullptr); // accepts hex floats
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    try {
        if (cmd == "fmt") { // fmt v...  -> one line per value: display|debug|json
            for (int i = 2; i < argc; ++i) {
                const double v = parse_double(argv[i]);
                std::cout << display(v) << "|" << debug(v) << "|" << json_number(v) << "\n";
            }
        } else if (cmd == "stats") { // stats v... -> compact json of Stats
            std::vector<double> v;
            for (int i = 2; i < argc; ++i) v.push_back(parse_double(argv[i]));
            std::cout << stats_json(Stats::compute(v), 0, false) << "\n";
        } else if (cmd == "output") { // output MODE psnr,ssim,msssim,ssimulacra2(0/1 flags) < lines of scores
            Output o;
            if (argc < 4 || !parse_output(argv[2], o)) return 2;
            Metrics m;
            m.psnr = argv[3][0] == '1'; m.ssim = argv[3][1] == '1'; m.msssim = argv[3][2] == '1'; m.ssimulacra2 = argv[3][3] == '1';
            std::vector<double> a, b, c, d;
            output_prepare(o, m, std::cout);
            std::string line;
            size_t n = 0;
            while (std::getline(std::cin, line)) {
                std::istringstream ss(line);
                FrameScores r;
                std::string t;
                if (m.psnr) { ss >> t; r.psnr = parse_double(t); a.push_back(*r.psnr); }
                if (m.ssim) { ss >> t; r.ssim = parse_double(t); b.push_back(*r.ssim); }
                if (m.msssim) { ss >> t; r.msssim = parse_double(t); c.push_back(*r.msssim); }
                if (m.ssimulacra2) { ss >> t; r.ssimulacra2 = parse_double(t); d.push_back(*r.ssimulacra2); }
                output_single_score(o, r, std::cout);
                ++n;
            }
            MetricsResults res;
            res.frame_count = n;
            if (m.psnr) res.psnr = MetricAggregate::from(a);
            if (m.ssim) res.ssim = MetricAggregate::from(b);
            if (m.msssim) res.msssim = MetricAggregate::from(c);
            if (m.ssimulacra2) res.ssimulacra2 = MetricAggregate::from(d);
            output_results(o, res, std::cout);
        } else if (cmd == "source") {
            // source PATH OUT [--width W --height H --bits B --cp N --mc N --tc N --skip N]: every frame's bytes as handed
            // to the engine are appended to OUT (biplanar kinds: `rows` luma rows + chroma rows at the surface pitch);
            // stdout: one description line, then one line per frame
            SourceHints h;
            uint32_t skip = 0;
            int readahead = -1, lookahead = -1;
            for (int i = 4; i + 1 < argc; i += 2) {
                const std::string k = argv[i];
                const int v = atoi(argv[i + 1]);
                if (k == "--width") h.width = v; else if (k == "--height") h.height = v; else if (k == "--bits") h.bits = v;
                else if (k == "--cp") h.cp = v; else if (k == "--mc") h.mc = v; else if (k == "--tc") h.tc = v; else if (k == "--skip") skip = v;
                else if (k == "--readahead") readahead = v; else if (k == "--lookahead") lookahead = v;
            }
            auto src = create_source(argv[2], h);
            const auto cc = src->color_characteristics();
            std::cout << src->format_id().str() << " " << src->width() << " " << src->height() << " " << to_string(cc.first.cp) << " "
                      << to_string(cc.first.mc) << " " << to_string(cc.first.tc) << " " << to_string(cc.second) << " " << src->frame_count() << "\n";
            std::ofstream out(argv[3], std::ios::binary);
            if (readahead >= 0) src->set_readahead(readahead != 0);
            if (lookahead >= 0) src->set_lookahead((size_t)lookahead);
            src->skip_frames(skip);
            HwFrame f;
            while (src->next_frame(f)) {
                const uint32_t w = src->width(), hh = src->height();
                if (f.kind == HwFrame::NvDecNV12 || f.kind == HwFrame::NvDecP016) {
                    const size_t luma_rows = ((const char *)f.uv - (const char *)f.data) / f.pitch, crows = (hh + 1) / 2;
                    out.write((const char *)f.data, (std::streamsize)(f.pitch * luma_rows));
                    out.write((const char *)f.uv, (std::streamsize)(f.pitch * crows));
                    std::cout << (f.kind == HwFrame::NvDecNV12 ? "nv12 " : "p016 ") << f.pitch << " " << luma_rows << " " << crows << "\n";
                } else if (f.kind == HwFrame::Planar420P10) {
                    const size_t crows = (hh + 1) / 2;
                    out.write((const char *)f.data, (std::streamsize)(f.pitch * hh));
                    out.write((const char *)f.u, (std::streamsize)(f.pitch_uv * crows));
                    out.write((const char *)f.v, (std::streamsize)(f.pitch_uv * crows));
                    std::cout << "p10 " << f.bits << " " << f.pitch << " " << f.pitch_uv << " " << hh << " " << crows << "\n";
                } else if (f.kind == HwFrame::Planar420) {
                    const size_t crows = (hh + 1) / 2;
                    out.write((const char *)f.data, (std::streamsize)(f.pitch * hh));
                    out.write((const char *)f.u, (std::streamsize)(f.pitch_uv * crows));
                    out.write((const char *)f.v, (std::streamsize)(f.pitch_uv * crows));
                    std::cout << "i420 " << f.bits << " " << f.pitch << " " << f.pitch_uv << " " << hh << " " << crows << "\n";
                } else {
                    out.write((const char *)f.data, (std::streamsize)(f.pitch * hh));
                    std::cout << (f.kind == HwFrame::Npp8 ? "rgb8 " : f.kind == HwFrame::Npp16 ? "rgb16 " : "rgbf32 ") << f.pitch << " " << w << "\n";
                }
            }
        } else if (cmd == "readbench") { // readbench PATH [lookahead]: next_frame over the whole stream, nothing else -> pictures/s and GB/s of the reader alone
            SourceHints h;
            auto src = create_source(argv[2], h);
            if (argc > 3) src->set_lookahead((size_t)atoi(argv[3]));
            HwFrame f;
            size_t n = 0, bytes = 0;
            unsigned long long sum = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (src->next_frame(f)) {
                ++n;
                const size_t crows = (src->height() + 1) / 2;
                bytes += f.pitch * src->height() + 2 * f.pitch_uv * crows;
                sum += ((const unsigned char *)f.data)[0];
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::cout << n << " pictures in " << dt << " s = " << (double)n / dt << " pictures/s, " << (double)bytes / dt / 1e9 << " GB/s, pinned " << (f.pinned ? 1 : 0)
                      << " cpus " << effective_cpus() << " (" << sum << ")\n";
        } else if (cmd == "demux") { // demux PATH OUT: container codec frame_count, then OUT = [u32 len][bytes] of init() and of every demux() piece
            FILE *f = fopen(argv[2], "rb");
            if (!f) throw std::runtime_error("open");
            std::string why;
            auto dm = probe_video(f, why);
            if (!dm) { fclose(f); std::cout << "NONE " << why << "\n"; return 0; }
            std::ofstream out(argv[3], std::ios::binary);
            std::vector<uint8_t> pkt;
            auto put = [&](const std::vector<uint8_t> &v) { const uint32_t n = (uint32_t)v.size(); out.write((const char *)&n, 4); out.write((const char *)v.data(), n); };
            dm->init(pkt);
            put(pkt);
            size_t n = 0;
            auto parse_all = [&](const std::vector<uint8_t> &v) { // every header parser sees every piece (the fuzzer's way into them)
                if (v.empty()) return;
                (void)av1_parse_sequence_header(v.data(), v.size());
                (void)mpeg2_parse_sequence(v.data(), v.size());
                for (size_t i = 0; i + 4 < v.size(); ++i)
                    if (v[i] == 0 && v[i + 1] == 0 && v[i + 2] == 1) (void)h264_parse_sps(v.data() + i + 3, v.size() - i - 3);
            };
            parse_all(pkt);
            while (dm->demux(pkt)) { put(pkt); parse_all(pkt); ++n; }
            std::cout << to_string(dm->container()) << " " << to_string(dm->codec()) << " " << dm->frame_count() << " " << n << "\n";
        } else if (cmd == "seqhdr") { // seqhdr h264|av1|mpeg2 HEX
            std::vector<uint8_t> b;
            const std::string hex = argv[3];
            for (size_t i = 0; i + 1 < hex.size(); i += 2) b.push_back((uint8_t)std::stoul(hex.substr(i, 2), nullptr, 16));
            const std::string k = argv[2];
            const StreamFormat f = k == "h264" ? h264_parse_sps(b.data(), b.size()) : (k == "av1" ? av1_parse_sequence_header(b.data(), b.size()) : mpeg2_parse_sequence(b.data(), b.size()));
            std::cout << (f.valid ? 1 : 0) << " " << f.width << " " << f.height << " " << f.bit_depth << " " << f.chroma_format << " " << f.cp << " " << f.mc << " "
                      << f.tc << " " << (f.full_range ? 1 : 0) << "\n";
        } else if (cmd == "colors") { // colors CP MC TC HEIGHT -> resolved characteristics + engine codes (or the error)
            const ColorCharacteristics c = ColorCharacteristics::from_codes(atoi(argv[2]), atoi(argv[3]), atoi(argv[4])).or_(color_characteristics_fallback(atoi(argv[5])));
            std::cout << to_string(c.cp) << " " << to_string(c.mc) << " " << to_string(c.tc) << " ";
            std::cout << get_color_matrix(c) << " " << get_transfer(c) << "\n";
        } else if (cmd == "ranks") {
            // ranks WORLD N_INDICES EVERY [FAIL_RANK [HANG_RANK]]: the launcher and the pipe transport of `turbo-metrics --ranks N` without a GPU --
            // this program starts WORLD copies of itself; rank r "scores" the selected decode indices of shard_range(r) with made-up values,
            // ONE reduce brings them to rank 0, which prints them in decode order (hex floats).  FAIL_RANK exits with code 3 before the
            // reduce, HANG_RANK never gets there (the launcher must end it).
            const int world = atoi(argv[2]);
            const uint32_t total = (uint32_t)atoi(argv[3]), every = (uint32_t)atoi(argv[4]);
            const int fail_rank = argc > 5 ? atoi(argv[5]) : -1, hang_rank = argc > 6 ? atoi(argv[6]) : -1;
            RankEnv env;
            if (!rank_env(env)) { if (getenv("TM_TEST_PRINT_PID")) { printf("launcher %d\n", (int)getpid()); fflush(stdout); } return launch_ranks(argv, world, 60.0); }
            if (env.rank == fail_rank) return 3;
            if (env.rank == hang_rank) { for (;;) std::this_thread::sleep_for(std::chrono::seconds(1)); }
            Metrics m; m.psnr = true; m.ssimulacra2 = true;
            ScoreVector sv(m, total);
            uint32_t lo = 0, hi = 0;
            shard_range(total, (uint32_t)env.rank, (uint32_t)env.world, lo, hi);
            for (uint32_t dc = lo; dc < hi; ++dc) {
                if (every > 1 && dc != 0 && dc % every != 0) continue;
                FrameScores f;
                f.psnr = dc == 5 ? INFINITY : 30.0 + 0.1 * dc;
                f.ssimulacra2 = 100.0 / (1.0 + dc) - 7.0;
                sv.put(dc, f);
            }
            sv.add_decoded(hi - lo);
            auto tr = make_rank_transport(env, "pipe");
            try { tr->reduce_sum_to_root(sv.v); }
            catch (const RankPeerLost &) { return RANK_PEER_LOST; }
            if (env.rank == 0) {
                std::cout << "lo " << lo << " hi " << hi << " decoded " << sv.decoded() << " " << tr->name() << "\n";
                for (const FrameScores &f : sv.frames()) printf("%a %a\n", *f.psnr, *f.ssimulacra2);
            } else std::cout << "rank " << env.rank << " must not be heard on stdout\n";
        } else if (cmd == "shard") { // shard N WORLD -> "lo hi" per rank
            for (uint32_t r = 0; r < (uint32_t)atoi(argv[3]); ++r) {
                uint32_t lo, hi;
                shard_range((uint32_t)atoi(argv[2]), r, (uint32_t)atoi(argv[3]), lo, hi);
                std::cout << lo << " " << hi << "\n";
            }
        } else return 2;
    } catch (const std::exception &e) {
        std::cout << "ERROR: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
