export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
cat > /tmp/run_x.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from tm_pkg import tm
tm.init_hip(0)
w, h, B = 1920, 1080, 32
eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
eng.set_variant(768 + 9)
keep = []
for n in range(4):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    keep.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
for s in range(B):
    (rt, rp, rch), (dt, dp, dch) = keep[s % 4]
    eng.set_pair(s, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
for _ in range(3):
    eng.compute_async(); eng.sync()
PY
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01x_pmc_fetch -- python3 /tmp/run_x.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r01x_pmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:50]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "tmk" in k: print(k, len(v), "R GB (x2 corrected):", round(2 * sum(v[1:]) / max(1, len(v) - 1) * 1024 / 1e9, 3))
PY
