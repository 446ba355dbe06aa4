#!/bin/bash
# GPU box: SQ-level counters per kernel (one pass, <= 8 SQ counters).  usage: pmc_sq.sh TAG [bench args]
set -u
TAG=${1:-sq}
shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --pmc ${TM_SQ_COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU} --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-extras --detail-file /tmp/tm_sq_detail.json "$@" > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
python3 - <<PY | tee gpurun_out/${TAG}_sq_summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "tmk" not in k or ", 1>" in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s %.4g" % (c, sum(v) / len(v)))
    g = lambda n: sum(d[n]) / len(d[n]) if d.get(n) else 0.0
    if g("SQ_WAVES"): print("   VALU instructions per wave     %.0f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
    if g("SQ_BUSY_CYCLES"): print("   VALU active / busy cycles       %.2f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES")))
PY
