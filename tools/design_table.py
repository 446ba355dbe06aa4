#!/usr/bin/env python3
"""Rewrites the measured block of docs/LABBOOK.md section 5 (between the bench-table markers) from profiles/<TAG>_*: the table of the
default bench line and the rocprofv3 cross-check.  usage: design_table.py TAG"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = sys.argv[1]
P = lambda n: os.path.join(ROOT, "profiles", n)
wall = [l for l in open(P(f"{T}_bench.json")) if l.startswith("real")]
d = json.load(open(P(f"{T}_bench_detail.json")))  # the full record of the default run (the stdout line is the compact one)
prof = json.load(open(P(f"{T}_prof_bench_1080p_detail.json")))
class Stats(dict):
    """kernel name -> (calls, average ms); a name that is not there is looked up by prefix (template arguments added by later rounds)"""
    def __missing__(self, key):
        stem = key.rstrip(">")
        hits = [k for k in self if k.startswith(stem)]
        if not hits:
            raise KeyError(key)
        return self[sorted(hits, key=len)[0]]
def stats(name):
    out = Stats()
    for r in csv.DictReader(open(P(name))):
        out[r["Name"].replace("void ", "").split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    return out
fmt = lambda v: f"{v:,.0f}".replace(",", " ")
def row(label, x, bold=True, r01=None):
    k = x["kernels"]
    ss = f'{k["k_ssim_stage"]["avg_launch_ms"]:.2f}' if "k_ssim_stage" in k else "—"
    val = f"**{fmt(x['value'])}**" if bold else fmt(x["value"])
    if r01: val += f" (r01: {r01})"
    st = x["stages"]
    al = x.get("kernels_alone", {}).get("kernels")
    ef = f'{k["k_blur_edge_fused"]["avg_launch_ms"]:.2f}' if "k_blur_edge_fused" in k else "—"
    def pair(name):  # beside the fused kernel (as the step runs) / alone on the chip
        return f'{k[name]["avg_launch_ms"]:.2f}' + (f' ({al[name]["avg_launch_ms"]:.2f})' if al else "")
    frac = f"{k['k_blur_v_jobs']['frac']:.3f}" + (f" ({al['k_blur_v_jobs']['frac']:.3f})" if al else "")
    if al: ef += f' ({al["k_blur_edge_fused"]["avg_launch_ms"]:.2f})'
    return (f"| {label} | {val} | {x['ms_per_step']:.2f} | {k['k_ingest_rows']['avg_launch_ms']:.2f} | {pair('k_blur_v_jobs')} | "
            f"{pair('k_blur_h_jobs_x')} | {ef} | {ss} | {'**' if bold and not r01 else ''}{frac}{'**' if bold and not r01 else ''} | "
            f"{st['blur_reduce_stage_frac']:.3f} ({st['survey_8d_model_frac']:.3f} on the §8d model) |")
c = d["compare"]
w = d["workloads"]
rows = [
    f"| workload (1 GPU, inputs resident in HBM), `{T}_bench_detail.json`; in brackets: every kernel alone on the chip (`kernels_alone`) | pairs/s | ms/step | ingest | column pass (FULL jobs) | row pass (FULL jobs) | fused kernel of the EDGE jobs, beside the passes | SSIM stage | column-pass frac of 8 TB/s | blur+reduce stage frac (= `roofline.frac`: the three kernels as one concurrent group) |",
    "|---|---|---|---|---|---|---|---|---|---|",
    row("1080p NV12, SSIMULACRA2, 64 pairs/step (headline; drivers' runs: r01 11 384, r02 11 033, r03 13 911)", d),
    row(f"— the same command under rocprofv3, `{T}_prof_bench_1080p_detail.json`", prof),
    f"| — with all 108 sums (`compare`: every job FULL, no fused kernel) | {fmt(c['value'])} | {c['ms_per_step']:.2f} | {c['stage_ms']['ingest']:.2f} | {c['stage_ms']['blur_v']:.2f} | {c['stage_ms']['blur_h']:.2f} | — | — | "
    f"{14858e6 * 64 / 64 / c['stage_ms']['blur_v'] / 8e9 * 1e3 / 1e3:.2f} | {c['blur_reduce_stage_frac']:.3f} |",
    row("4K P016, SSIMULACRA2, 24 pairs/step (r02 driver: 2 822; r03: 3 500)", w["4k_p016"]),
    row("1080p PSNR + MS-SSIM + SSIMULACRA2 in one pass", w["1080p_nv12_fused"], r01="6 545, r02: 8 292, r03: 9 770"),
    row("4K PSNR + MS-SSIM + SSIMULACRA2 in one pass", w["4k_p016_fused"], r01="1 633, r02: 2 059, r03: 2 453"),
]
s1, sf, s4f = stats(f"{T}_kernel_stats_1080p_b128.csv"), stats(f"{T}_kernel_stats_1080p_b128_fused.csv"), stats(f"{T}_kernel_stats_4k_b48_fused.csv")
pk = prof["kernels"]
hf, cb, fx = d["host_fed"], d["cpu_baseline"], d["fixed_stream"]
bc = d["batch_curve"]["points"]
cli = d["cli_end_to_end"]
curve = " · ".join(f"{p_['batch']}: {fmt(p_['value'])} ({p_['ms_per_step']:.2f} ms, {p_['engine_mem_GB']} GB)" for p_ in bc)
ing = [k for k in s1 if "k_ingest_rows" in k][0]
efk = [k for k in s1 if "k_blur_edge_fused" in k][0]
text = "\n".join(rows) + f"""

(`python bench.py`, {wall[0].split()[1] if wall else '?'} wall; `fixed_stream` {fmt(fx['value'])} pairs/s over 2 048 pairs, {fmt(fx['long']['value'])} over 16 384; `host_fed` {fmt(hf['1080p_nv12']['value'])} pairs/s at 1080p =
{hf['1080p_nv12']['h2d_GBs_per_gpu']:.1f} GB/s over PCIe, {fmt(hf['4k_p016']['value'])} at 4K = {hf['4k_p016']['h2d_GBs_per_gpu']:.1f} GB/s (40 steps each); `cpu_baseline` {cb['value']:.2f} pairs/s on one core, {cb['all_cores']['value']:.1f} on the {cb['all_cores']['cores']} CPUs the container may use ({cb['host_cpus']} visible).)
`batch_curve` (1080p, pairs per launch: pairs/s (ms per step, engine memory)): {curve}.
`cli_end_to_end` (the C++ binary on Y4M clips in tmpfs, its own "Processed" figure): 1080p 8-bit {fmt(cli['1080p_yuv420p']['default']['pairs_per_s'])} pairs/s with the defaults (`--batch` by picture size: 16) and {fmt(cli['1080p_yuv420p']['batch16']['pairs_per_s'])} with `--batch 16`;
4K 10-bit {fmt(cli['4k_yuv420p10']['default']['pairs_per_s'])} (batch 4) / {fmt(cli['4k_yuv420p10']['batch16']['pairs_per_s'])}.
rocprofv3 of the same command (`profiles/{T}_kernel_stats_1080p_b128.csv`): `k_blur_v_jobs<32,16,0>` {s1['tmk::k_blur_v_jobs<32, 16, 0>'][1]:.3f} ms average over {s1['tmk::k_blur_v_jobs<32, 16, 0>'][0]}
launches vs {pk['k_blur_v_jobs']['avg_launch_ms']:.3f} ms from the HIP events of the timed steps of that run; row pass {s1['tmk::k_blur_h_jobs_x<16, 8, 32, 16, 0>'][1]:.3f} vs {pk['k_blur_h_jobs_x']['avg_launch_ms']:.3f}; `k_ingest_rows` {s1[ing][1]:.3f} +
`k_ingest_upper_rd` {s1['tmk::k_ingest_upper_rd'][1]:.3f} vs {pk['k_ingest_rows']['avg_launch_ms']:.3f} for the stage; `{efk.replace('tmk::', '')}` {s1[efk][1]:.3f} vs {pk['k_blur_edge_fused']['avg_launch_ms']:.3f} (+ `k_finish_edge`
{s1['tmk::k_finish_edge'][1]:.3f}); the launches of the placement search are listed apart as `…, 1>`. With the SSIM stage
(`{T}_kernel_stats_1080p_b128_fused.csv`): `k_ssim_stream` {sf['tmk::k_ssim_stream'][1]:.3f}, `k_ssim_pyramid` {sf['tmk::k_ssim_pyramid'][1]:.3f} (4K, `{T}_kernel_stats_4k_b48_fused.csv`:
{s4f['tmk::k_ssim_stream'][1]:.3f} and {s4f['tmk::k_ssim_pyramid'][1]:.3f})."""
if os.path.exists(P(f"{T}_kernel_stats_1080p_b128_alone.csv")):
    sa = stats(f"{T}_kernel_stats_1080p_b128_alone.csv")
    efa = [k for k in sa if "k_blur_edge_fused" in k][0]
    al = d["kernels_alone"]["kernels"]
    text += f"""
Every kernel alone on the chip (`bench.py --edge-beside 0`: the fused kernel behind the row pass; `profiles/{T}_kernel_stats_1080p_b128_alone.csv`, rocprofv3 of
`bench.py --no-extras` in that mode): `k_blur_v_jobs<32,16,0>` {sa['tmk::k_blur_v_jobs<32, 16, 0>'][1]:.3f} ms = {7424901120 / sa['tmk::k_blur_v_jobs<32, 16, 0>'][1] / 8e9 * 1e3 / 1e3:.3f} of 8 TB/s on its 7.42 GB (`kernels_alone` of the bench line: {al['k_blur_v_jobs']['avg_launch_ms']:.3f} ms,
{al['k_blur_v_jobs']['frac']:.3f}), row pass {sa['tmk::k_blur_h_jobs_x<16, 8, 32, 16, 0>'][1]:.3f} ms = {7424901120 / sa['tmk::k_blur_h_jobs_x<16, 8, 32, 16, 0>'][1] / 8e9 * 1e3 / 1e3:.3f} ({al['k_blur_h_jobs_x']['avg_launch_ms']:.3f}, {al['k_blur_h_jobs_x']['frac']:.3f}), `{efa.replace('tmk::', '')}` {sa[efa][1]:.3f} ms ({al['k_blur_edge_fused']['avg_launch_ms']:.3f})."""
p = os.path.join(ROOT, "docs", "LABBOOK.md")
s = open(p).read()
a, b = "<!-- bench-table:begin -->", "<!-- bench-table:end -->"
i, j = s.index(a) + len(a), s.index(b)
s = s[:i] + "\n" + text + "\n" + s[j:]
s = re.sub(r"profiles/r0[0-9][a-z]_sq_counters_\*", f"profiles/{T}_sq_counters_*", s)
open(p, "w").write(s)
print(text)
