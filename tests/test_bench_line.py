"""CPU tier: the ONE line bench.py prints must stay readable by the driver (BENCH_r03.json came back with
`parsed: null` because the line had grown to 22 KB).  compact_line() is fed the full record of that very run
(tests/golden/bench_detail_r03.json = the 22-KB line) and must give <= 4 KB that round-trip through json with every key
the contract, the roofline and the CPU baseline need."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _detail():
    with open(os.path.join(ROOT, "tests", "golden", "bench_detail_r03.json")) as f:
        return json.load(f)


def test_compact_line_is_short_and_complete():
    import bench
    d = _detail()
    assert len(json.dumps(d)) > 20000  # the record that broke the driver's parse
    d["detail_file"] = "bench_detail.json"
    line = bench.compact_line(d)
    assert "\n" not in line and len(line) < 4096, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
        assert k in out, k
    for k in ("workload", "baseline_config", "width", "height", "input", "pairs_per_step_per_gpu", "full_sums", "parallelism"):
        assert k in out["config"], k
    assert "model" not in out["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
              "bytes_model", "frac_alone", "full_sums_frac"):
        assert k in out["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "all_cores"):
        assert k in out["cpu_baseline"], k
    assert out["value"] == round(d["value"], 4) and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-4
    sm = out["summary"]
    assert set(("4k_p016", "1080p_nv12_fused", "4k_p016_fused", "fixed_stream", "fixed_stream_long", "host_fed", "batch_curve", "cli_end_to_end")) <= set(sm)
    assert all(not isinstance(v, dict) or all(not isinstance(x, dict) or tag == "cli_end_to_end" for x in v.values()) for tag, v in sm.items())
    assert sm["fixed_stream_long"]["pairs"] == 16384 and sm["fixed_stream"]["sha"] == d["fixed_stream"]["scores_sha256_16"]
    assert [b for b, _ in sm["batch_curve"]] == [1, 2, 4, 8, 16, 32, 64]


def test_compact_line_survives_missing_legs_and_oversized_strings():
    import bench
    d = _detail()
    for k in ("workloads", "host_fed", "batch_curve", "cli_end_to_end", "cpu_baseline", "compare", "fixed_stream"):
        d.pop(k)
    out = json.loads(bench.compact_line(d))
    assert "cpu_baseline" not in out and "full_sums_frac" not in out["roofline"] and out["value"] > 0
    d = _detail()
    d["cli_end_to_end"] = {("clip%d" % i): {"error": "x" * 500} for i in range(40)}
    d["roofline"]["bytes_model"] = "y" * 3000
    d["cpu_baseline"]["sample"] = "z" * 3000
    line = bench.compact_line(d)
    assert len(line) <= bench.LINE_LIMIT
    assert json.loads(line)["roofline"]["frac"] > 0


def test_non_finite_numbers_never_reach_the_line():
    import bench
    d = _detail()
    d["score_mean"] = float("nan")
    d["roofline"]["traffic"] = None
    d["host_fed"]["4k_p016"]["value"] = float("inf")
    line = bench.compact_line(d)
    assert "NaN" not in line and "Infinity" not in line
    json.loads(line)
