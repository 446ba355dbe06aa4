#!/bin/bash
# GPU box: the round's evidence in one call.  Order matters: the PMC traffic passes come first and are copied over the
# profiles/pmc_traffic_*.json of the snapshot, so that the default bench line that follows finds a traffic profile stamped with
# the kernel sources it runs (bench.py refuses another version's).  Then GPU tests, smoke, the default bench line (headline +
# workloads + fixed stream + host-fed + cpu baseline), the 2-rank bench over gloo, rocprofv3 kernel stats (SSIMULACRA2 alone
# and fused, 1080p and 4K), SQ counters.  Outputs -> gpurun_out/<TAG>_*; tools/collect_profiles.sh copies what is judged.
set -u
TAG=${1:-r06z}
bash tools/pmc_traffic.sh $TAG 1080p_nv12 > /dev/null
bash tools/pmc_traffic.sh $TAG 1080p_nv12 --full-sums > /dev/null
bash tools/pmc_traffic.sh $TAG 4k_p016 > /dev/null
cp gpurun_out/${TAG}_pmc_traffic_1080p_nv12.json profiles/pmc_traffic_1080p_nv12_b128.json
cp gpurun_out/${TAG}_pmc_traffic_1080p_nv12_full.json profiles/pmc_traffic_1080p_nv12_b128_full.json
cp gpurun_out/${TAG}_pmc_traffic_4k_p016.json profiles/pmc_traffic_4k_p016_b48.json
bash tools/gpu_check.sh $TAG
bash tools/gpu_prof.sh ${TAG}_fused --metrics psnr,msssim,ssimulacra2
bash tools/gpu_prof.sh ${TAG}_alone --edge-beside 0   # every kernel alone on the chip: the fused kernel of the EDGE jobs behind the row pass
bash tools/gpu_prof.sh ${TAG}_4k --workload 4k_p016
bash tools/gpu_prof.sh ${TAG}_fused4k --workload 4k_p016 --metrics psnr,msssim,ssimulacra2
bash tools/pmc_sq.sh $TAG > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_fused --metrics psnr,msssim,ssimulacra2 > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}_b1 --batch 1 --settle-ms 30 > /dev/null 2>&1     # one pair per launch: what the multi-wave row pass is made of
# round 4: small launches, two batches in flight, the CLI and its reader alone, create / destroy soak (no torch in that process)
timeout 400 python tools/small_launch_probe.py 1,2,4,8,16,32 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_small_launch_probe.log
for a in "1080p_nv12 64 60" "1080p_nv12 32 100" "4k_p016 24 40" "1080p_nv12 64 40 psnr,msssim,ssimulacra2"; do timeout 300 python tools/pipeline_probe.py $a 2>&1 | tail -1; done > gpurun_out/${TAG}_pipeline_probe.log
timeout 600 python tools/deferred_depth_probe.py 1,2,3,4,6,8 > gpurun_out/${TAG}_deferred_depth.log 2>&1
timeout 300 python tests/soak/create_destroy_soak.py 200 > gpurun_out/${TAG}_soak_create_destroy.log 2>&1
timeout 300 python tools/cli_bench.py --size 1080p 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_cli_1080p.log
timeout 300 python tools/cli_bench.py --size 4k 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_cli_4k.log
(bash tools/cli_ab.sh 1080p 3 "" "--tune 7=0" "--tune 7=0 --tune 8=1"; bash tools/cli_ab.sh 4k 3 "" "--tune 7=0" "--tune 7=0 --tune 8=1") 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_cli_ab.log
(timeout 300 python3 tools/host_fed_ab.py 1080p 2; timeout 300 python3 tools/host_fed_ab.py 4k 2) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_host_fed_ab.log
# round 6: the write ceiling of the part, the ingest fold A/B on this tree (laboratory library, TM_VARIANT_UPPER_KERNEL = round 5's arrangement)
timeout 300 tools/microbench/wr_ceiling > gpurun_out/${TAG}_wr_ceiling.log 2>&1
LIB=turbo-metrics_amd/lab/libturbometrics_hip_lab.so
(for B in 128 64 1; do echo "== batch $B"; timeout 900 python tools/lib_ab.py fold=$LIB upper=$LIB@0x2000 --rounds 3 --batch $B 2>&1 | tail -9; done; echo "== 4k"; timeout 900 python tools/lib_ab.py fold=$LIB upper=$LIB@0x2000 --rounds 2 --workload 4k_p016 2>&1 | tail -7) > gpurun_out/${TAG}_fold_ab.log 2>&1
ls gpurun_out | grep $TAG | head -60
