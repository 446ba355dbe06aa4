#!/bin/bash
# GPU box: HBM traffic of each kernel from the TCC fabric counters, one counter per rocprofv3 pass
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# The result is stamped with the content hash of csrc/ (tools/parse_pmc.py): bench.py only reports it for that version.
set -u
TAG=${1:-r02}
WL=${2:-1080p_nv12}
EXTRA=${3:-}          # e.g. --full-sums
SUF=${EXTRA:+_full}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_${WL}${SUF}_$C -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-extras $EXTRA > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
done
cd $R
python3 tools/parse_pmc.py gpurun_out/${TAG}_pmc_${WL}${SUF}_FETCH_SIZE gpurun_out/${TAG}_pmc_${WL}${SUF}_WRITE_SIZE $WL > gpurun_out/${TAG}_pmc_traffic_$WL$SUF.json
cat gpurun_out/${TAG}_pmc_traffic_$WL$SUF.json
