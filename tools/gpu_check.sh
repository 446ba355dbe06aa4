#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench (default: the compact line on stdout, the full record in <TAG>_bench_detail.json),
# the 2-rank bench over gloo, rocprof kernel-trace. Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02}
echo "== rocminfo" > gpurun_out/${TAG}_env.log
(rocminfo | grep -E "Marketing Name|gfx" | head -6; nproc; lscpu | grep "Model name") >> gpurun_out/${TAG}_env.log 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/${TAG}_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/${TAG}_smoke.log
echo "== bench (default)"
( time timeout 900 python bench.py --detail-file gpurun_out/${TAG}_bench_detail.json ) 2>&1 | tail -8 | tee gpurun_out/${TAG}_bench.json
echo "== bench --gpus 2 over gloo on one device"
TM_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 --no-extras --detail-file gpurun_out/${TAG}_bench_2ranks_gloo_detail.json 2>&1 | tail -3 | tee gpurun_out/${TAG}_bench_2ranks_gloo.json
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-compare --no-extras --detail-file $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench_detail.json > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/${TAG}_prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
