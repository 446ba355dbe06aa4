#!/bin/bash
# GPU box: the hardware-queue lanes of the engine's streams (tm_engine.hip) on and off -- where the streams land, what an engine's creation
# costs, the headline with an RCCL communicator created BEFORE the engine (bench.py --gpus N does that), the CLI's loops
export TMPDIR=/tmp
for q in 1 0; do
TM_QUEUE_LANES=$q python - <<'PY' 2>&1 | grep -v "candidate\|amdgpu"
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from tm_pkg import tm
tm.init_hip(0); tm.ffi.lib().tm_set_debug_log(1); tm.set_placement_candidates(1)
print("TM_QUEUE_LANES =", os.environ.get("TM_QUEUE_LANES"), flush=True)
t = []
keep = []
for b in (16, 16, 1, 1, 1, 1):
    t0 = time.perf_counter(); keep.append(tm.TurboMetrics(1920, 1080, tm.Metrics(ssimulacra2=True), batch=b)); t.append(round((time.perf_counter() - t0) * 1e3, 1))
print("engine creation ms (batch 16, 16, 1, 1, 1, 1):", t, flush=True)
PY
done
for f in 1; do for q in 1 0; do echo "== TM_BENCH_FORCE_DIST=$f TM_QUEUE_LANES=$q"; RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 TM_QUEUE_LANES=$q TM_BENCH_FORCE_DIST=$f timeout 300 python bench.py --no-extras --no-cpu-baseline --no-compare --steps 40 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"roofline\"][\"frac\"], d[\"summary\"].get(\"stage_ms\"))"; done; done
bash tools/cli_ab.sh 1080p 3 "" "TM_QUEUE_LANES=0" "--loop deferred --in-flight 3" "--loop deferred --in-flight 3 TM_QUEUE_LANES=0" "--loop deferred --in-flight 4" "--loop deferred --in-flight 4 TM_QUEUE_LANES=0" 2>&1 | cut -c1-100
