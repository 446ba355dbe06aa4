#!/bin/bash
# round 6: GPU test tier, then A/B of the ingest fold (same library, TM_VARIANT_UPPER_KERNEL = the round-5 arrangement) at 128 / 64 / 16 / 1 pairs and 4K
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r06c}
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/${TAG}_pytest_gpu.log
LIB=turbo-metrics_amd/lab/libturbometrics_hip_lab.so
for B in 128 64 16 1; do
  echo "== fold A/B 1080p batch $B"
  timeout 900 python tools/lib_ab.py fold=$LIB upper=$LIB@0x2000 --rounds 3 --batch $B 2>&1 | tail -12
done | tee gpurun_out/${TAG}_fold_ab.log
echo "== fold A/B 4K"
timeout 900 python tools/lib_ab.py fold=$LIB upper=$LIB@0x2000 --rounds 2 --workload 4k_p016 2>&1 | tail -9 | tee -a gpurun_out/${TAG}_fold_ab.log
