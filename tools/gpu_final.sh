#!/bin/bash
# GPU box: the round's evidence in one call -- GPU tests, smoke, the default bench line (headline + workloads + fixed stream +
# host-fed + cpu baseline), the 2-rank bench over gloo, rocprofv3 kernel stats (SSIMULACRA2 alone and fused, 1080p and 4K),
# PMC traffic passes.  Outputs -> gpurun_out/<TAG>_*; copy what is to be judged into profiles/.
set -u
TAG=${1:-r02z}
bash tools/gpu_check.sh $TAG
bash tools/gpu_prof.sh ${TAG}_fused --metrics psnr,msssim,ssimulacra2
bash tools/gpu_prof.sh ${TAG}_4k --workload 4k_p016
bash tools/gpu_prof.sh ${TAG}_fused4k --workload 4k_p016 --metrics psnr,msssim,ssimulacra2
bash tools/pmc_traffic.sh $TAG 1080p_nv12 > /dev/null
bash tools/pmc_traffic.sh $TAG 1080p_nv12 --full-sums > /dev/null
bash tools/pmc_traffic.sh $TAG 4k_p016 > /dev/null
ls gpurun_out | grep $TAG | head -40
