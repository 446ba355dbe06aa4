#!/bin/bash
# GPU box: rocprofv3 kernel-trace of one bench.py configuration.  usage: gpu_prof.sh TAG [bench args]
set -u
TAG=${1:-prof}; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare --no-extras --detail-file $R/gpurun_out/${TAG}_prof_detail.json "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R
grep '^{"metric"' gpurun_out/${TAG}_prof.log | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))
except Exception as e: print('no json', e)
"
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/${TAG}_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tmk" in r["Name"]: print("  %-60s calls %4s  avg %.4f ms" % (r["Name"].replace("void ","")[:60], r["Calls"], float(r["AverageNs"])/1e6))
PY
