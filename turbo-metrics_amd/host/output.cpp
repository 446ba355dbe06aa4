// output.cpp -- see output.hpp
#include "output.hpp"

#include <sstream>
#include <vector>

#include "rust_fmt.hpp"

namespace tm_host {

bool parse_output(const std::string &s, Output &out)
{
    if (s == "default") out = Output::Default;
    else if (s == "json") out = Output::Json;
    else if (s == "json-lines") out = Output::JsonLines;
    else if (s == "csv") out = Output::CSV;
    else return false;
    return true;
}

namespace {

struct Named { const char *name; double value; };

std::vector<Named> stat_fields(const Stats &s)
{
    return {{"min", s.min}, {"max", s.max}, {"mean", s.mean}, {"var", s.var}, {"sample_var", s.sample_var}, {"stddev", s.stddev},
            {"sample_stddev", s.sample_stddev}, {"p1", s.p1}, {"p5", s.p5}, {"p50", s.p50}, {"p95", s.p95}, {"p99", s.p99}};
}

void csv_header(bool psnr, bool ssim, bool msssim, bool ssimu, std::ostream &os)
{
    bool first = true;
    auto put = [&](bool on, const char *n) { if (on) { os << (first ? "" : ",") << n; first = false; } };
    put(psnr, "psnr"); put(ssim, "ssim"); put(msssim, "msssim"); put(ssimu, "ssimulacra2");
    if (first) os << "\"\""; // csv::Writer writes an empty record as ""
    os << "\n";
}

void csv_row(const std::optional<double> &a, const std::optional<double> &b, const std::optional<double> &c, const std::optional<double> &d,
             std::ostream &os)
{
    bool first = true;
    auto put = [&](const std::optional<double> &v) { if (v) { os << (first ? "" : ",") << display(*v); first = false; } };
    put(a); put(b); put(c); put(d);
    if (first) os << "\"\"";
    os << "\n";
}

std::string frame_scores_json(const FrameScores &r)
{
    std::string s = "{";
    bool first = true;
    auto put = [&](const char *n, const std::optional<double> &v) {
        if (!v) return; // skip_serializing_if = "Option::is_none"
        s += (first ? "\"" : ",\"") + std::string(n) + "\":" + json_number(*v);
        first = false;
    };
    put("psnr", r.psnr); put("ssim", r.ssim); put("msssim", r.msssim); put("ssimulacra2", r.ssimulacra2);
    return s + "}";
}

} // namespace

std::string stats_debug_pretty(const Stats &s)
{
    std::string out = "Stats {\n";
    for (const Named &f : stat_fields(s)) out += "    " + std::string(f.name) + ": " + debug(f.value) + ",\n";
    return out + "}";
}

std::string stats_json(const Stats &s, int indent, bool pretty)
{
    std::string out = "{";
    const std::string pad((size_t)indent + 2, ' ');
    bool first = true;
    for (const Named &f : stat_fields(s)) {
        if (!first) out += ",";
        if (pretty) out += "\n" + pad;
        out += "\"" + std::string(f.name) + "\":" + (pretty ? " " : "") + json_number(f.value);
        first = false;
    }
    if (pretty) out += "\n" + std::string((size_t)indent, ' ');
    return out + "}";
}

void output_prepare(Output o, const Metrics &m, std::ostream &os)
{
    if (o == Output::CSV) csv_header(m.psnr, m.ssim, m.msssim, m.ssimulacra2, os);
}

void output_single_score(Output o, const FrameScores &r, std::ostream &os)
{
    if (o == Output::JsonLines) os << frame_scores_json(r) << "\n";
    else if (o == Output::CSV) csv_row(r.psnr, r.ssim, r.msssim, r.ssimulacra2, os);
}

void output_results(Output o, const MetricsResults &r, std::ostream &os)
{
    switch (o) {
    case Output::Default:
        if (r.psnr) os << "PSNR: " << stats_debug_pretty(r.psnr->stats) << "\n";
        if (r.ssim) os << "SSIM: " << stats_debug_pretty(r.ssim->stats) << "\n";
        if (r.msssim) os << "MSSSIM: " << stats_debug_pretty(r.msssim->stats) << "\n";
        if (r.ssimulacra2) os << "SSIMULACRA2: " << stats_debug_pretty(r.ssimulacra2->stats) << "\n";
        break;
    case Output::Json: { // serde_json::to_string_pretty: two-space indent, `"key": value`
        os << "{\n  \"frame_count\": " << r.frame_count;
        auto put = [&](const char *n, const std::optional<MetricAggregate> &a) {
            if (!a) return;
            os << ",\n  \"" << n << "\": {\n    \"scores\": [";
            for (size_t i = 0; i < a->scores.size(); ++i) os << (i ? ",\n      " : "\n      ") << json_number(a->scores[i]);
            os << (a->scores.empty() ? "]" : "\n    ]") << ",\n    \"stats\": " << stats_json(a->stats, 4, true) << "\n  }";
        };
        put("psnr", r.psnr); put("ssim", r.ssim); put("msssim", r.msssim); put("ssimulacra2", r.ssimulacra2);
        os << "\n}\n";
        break;
    }
    case Output::JsonLines: {
        const MetricsStats s = MetricsStats::from(r);
        os << "{\"frame_count\":" << s.frame_count;
        auto put = [&](const char *n, const std::optional<Stats> &st) { if (st) os << ",\"" << n << "\":" << stats_json(*st, 0, false); };
        put("psnr", s.psnr); put("ssim", s.ssim); put("msssim", s.msssim); put("ssimulacra2", s.ssimulacra2);
        os << "}\n";
        break;
    }
    case Output::CSV:
        csv_header((bool)r.psnr, (bool)r.ssim, (bool)r.msssim, (bool)r.ssimulacra2, os);
        for (size_t i = 0; i < r.frame_count; ++i) {
            auto at = [&](const std::optional<MetricAggregate> &a) { return a ? std::optional<double>(a->scores[i]) : std::nullopt; };
            csv_row(at(r.psnr), at(r.ssim), at(r.msssim), at(r.ssimulacra2), os);
        }
        break;
    }
}

} // namespace tm_host
