#!/bin/bash
# round 6: GPU test tier + the default bench line (host_fed 4k_p10, CLI 4K packed / words16)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r06b}
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/${TAG}_pytest_gpu.log
echo "== bench (default)"
( time timeout 900 python bench.py --detail-file gpurun_out/${TAG}_bench_detail.json ) 2>&1 | tail -6 | tee gpurun_out/${TAG}_bench.json
