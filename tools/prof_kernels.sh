#!/bin/bash
# GPU box: rocprofv3 kernel-trace of bench.py, prints mean duration per product kernel.  usage: prof_kernels.sh TAG [bench args]
set -u
TAG=${1:-prof}; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R
tail -1 gpurun_out/${TAG}_prof.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))
except Exception as e: print('no json', e)
"
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/${TAG}_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tmk" in r["Name"]: print("  %-44s calls %4s  avg %.4f ms" % (r["Name"].replace("void ","")[:44], r["Calls"], float(r["AverageNs"])/1e6))
PY
