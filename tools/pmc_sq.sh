#!/bin/bash
# GPU box: SQ-level counters per kernel (one pass, <= 8 SQ counters)
set -u
TAG=${1:-sq}
shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --pmc ${TM_SQ_COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU} --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not k.startswith("tmk") and "tmk" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s %.4g" % (c, sum(v) / len(v)))
PY
