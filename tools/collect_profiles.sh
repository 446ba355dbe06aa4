#!/bin/bash
# here (after gpurun merged gpurun_out/): copy the files of tools/gpu_final.sh <TAG> that are to be judged into profiles/
set -eu
T=${1:?tag}
latest() { ls -t "$1"/runc/*_kernel_stats.csv | head -1; }
cp gpurun_out/${T}_pmc_traffic_1080p_nv12.json profiles/pmc_traffic_1080p_nv12_b128.json
cp gpurun_out/${T}_pmc_traffic_1080p_nv12_full.json profiles/pmc_traffic_1080p_nv12_b128_full.json
cp gpurun_out/${T}_pmc_traffic_4k_p016.json profiles/pmc_traffic_4k_p016_b48.json
cp "$(latest gpurun_out/${T}_prof)" profiles/${T}_kernel_stats_1080p_b128.csv
cp "$(latest gpurun_out/${T}_fused_prof)" profiles/${T}_kernel_stats_1080p_b128_fused.csv
if [ -d gpurun_out/${T}_alone_prof ]; then cp "$(latest gpurun_out/${T}_alone_prof)" profiles/${T}_kernel_stats_1080p_b128_alone.csv; fi
cp "$(latest gpurun_out/${T}_4k_prof)" profiles/${T}_kernel_stats_4k_b48.csv
cp "$(latest gpurun_out/${T}_fused4k_prof)" profiles/${T}_kernel_stats_4k_b48_fused.csv
cp gpurun_out/${T}_prof_bench_detail.json profiles/${T}_prof_bench_1080p_detail.json
grep -h '^{"metric"' gpurun_out/${T}_bench_2ranks_gloo.json | tail -1 > profiles/${T}_bench_2ranks_gloo_one_device.json
cp gpurun_out/${T}_bench.json gpurun_out/${T}_bench_detail.json gpurun_out/${T}_pytest_gpu.log gpurun_out/${T}_smoke.log gpurun_out/${T}_env.log profiles/
cp gpurun_out/${T}_sq_summary.txt profiles/${T}_sq_counters_1080p_b128.txt
cp gpurun_out/${T}_fused_sq_summary.txt profiles/${T}_sq_counters_1080p_b128_fused.txt
cp gpurun_out/${T}_b1_sq_summary.txt profiles/${T}_sq_counters_1080p_b1.txt
for f in small_launch_probe deferred_depth pipeline_probe soak_create_destroy cli_1080p cli_4k cli_ab host_fed_ab wr_ceiling fold_ab; do cp gpurun_out/${T}_$f.log profiles/; done
python3 tools/trace_timeline.py "$(ls -t gpurun_out/${T}_prof/runc/*_kernel_trace.csv | head -1)" 2 > profiles/${T}_timeline_1080p_b128.txt
ls profiles | grep ${T}
