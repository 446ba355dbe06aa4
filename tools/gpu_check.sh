#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprof kernel-trace. Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
echo "== rocminfo" > gpurun_out/${TAG}_env.log
(rocminfo | grep -E "Marketing Name|gfx" | head -6; nproc; lscpu | grep "Model name") >> gpurun_out/${TAG}_env.log 2>&1
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/${TAG}_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/${TAG}_smoke.log
echo "== bench 1080p"
timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/${TAG}_bench_1080p.json
echo "== bench 4k"
timeout 600 python bench.py --workload 4k_p016 --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/${TAG}_bench_4k.json
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-compare > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/${TAG}_prof -name "*kernel_stats*.csv" | head -3
for f in $(find gpurun_out/${TAG}_prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
