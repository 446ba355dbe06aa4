#!/bin/bash
# GPU box: which hardware counter tells a slow placement of the pass-1 arena from a fast one?  The placement search itself is the
# experiment: it runs the column pass (k_blur_v_jobs<32,16,1>) on each of 8 candidate arenas in turns and, with tm_set_debug_log(1),
# prints each candidate's time; rocprofv3 --pmc gives the counters per dispatch.  One process per counter set.
set -u
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for SET in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmcpl_$i -- python3 $R/tools/placement_debug.py 1 > $R/gpurun_out/${TAG}_pmcpl_$i.log 2>&1
  grep -E "candidate|rror" $R/gpurun_out/${TAG}_pmcpl_$i.log
  python3 $R/tools/parse_pmc_placement.py $R/gpurun_out/${TAG}_pmcpl_$i
done
