// output.hpp -- the CLI's stdout formats, byte for byte those of crates/turbo-metrics-cli/src/output.rs:6-143:
//   Default    nothing per frame; `println!("PSNR: {:#?}", stats)` ... per selected metric at the end
//   Json       serde_json::to_string_pretty(MetricsResults) at the end
//   JsonLines  serde_json::to_string(FrameScores) per frame, then MetricsStats
//   CSV        header in prepare(), one row per frame, and -- as the reference does -- the header and every row AGAIN
//              in output_results() (output.rs:104-139)
// Numbers: CSV uses Rust `{}`; JSON uses serde_json (ryu); Default uses `{:#?}` (rust_fmt.hpp).
#pragma once
#include <ostream>
#include <string>

#include "turbo_metrics.hpp"

namespace tm_host {

enum class Output { Default, Json, JsonLines, CSV };

// clap ValueEnum names: default, json, json-lines, csv
bool parse_output(const std::string &s, Output &out);

void output_prepare(Output o, const Metrics &m, std::ostream &os);
void output_single_score(Output o, const FrameScores &r, std::ostream &os);
void output_results(Output o, const MetricsResults &r, std::ostream &os);

std::string stats_debug_pretty(const Stats &s);                // `{:#?}`
std::string stats_json(const Stats &s, int indent, bool pretty); // serde_json

} // namespace tm_host
