#!/bin/bash
# GPU box: the GPU test tier and the default bench line of the current tree (what the driver runs at round end) -> gpurun_out/<TAG>_*
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-quick}
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.log
( time timeout 900 python bench.py --detail-file gpurun_out/${TAG}_bench_detail.json ) 2>&1 | tail -6 | tee gpurun_out/${TAG}_bench.json
