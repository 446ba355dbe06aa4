#!/bin/bash
# round 6, first call: write-ceiling microbenchmark, GPU test tier, baseline bench line on this box
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r06a}
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null) > gpurun_out/${TAG}_env.log 2>&1
echo "== wr_ceiling"
timeout 300 tools/microbench/wr_ceiling 2>&1 | tee gpurun_out/${TAG}_wr_ceiling.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/${TAG}_pytest_gpu.log
echo "== bench (default)"
( time timeout 900 python bench.py --detail-file gpurun_out/${TAG}_bench_detail.json ) 2>&1 | tail -6 | tee gpurun_out/${TAG}_bench.json
