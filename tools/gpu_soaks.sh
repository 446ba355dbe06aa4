#!/bin/bash
# GPU box: the long runs of the soak programs (tests/soak/) on the current tree -> gpurun_out/<TAG>_long_soaks.log
set -u
TAG=${1:-r06s}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_long_soaks.log
: > $L
run() { timeout "$1" python "${@:2}" 2>&1 | grep -v amdgpu.ids | tail -4 >> $L; }
run 900 tests/soak/variant_sweep_soak.py 5000 1300
run 300 tests/soak/engine_state_soak.py 100000 416 240
run 300 tests/soak/engine_state_soak.py 20000 1920 1080
run 900 tests/soak/surface_sweep_soak.py 8000
run 600 tests/soak/random_sweep_soak.py 20000 8000
run 900 tests/soak/cli_sweep_soak.py 100
run 300 tests/soak/abi_fuzz_soak.py 20000
run 300 tests/soak/deferred_soak.py 20000
run 300 tests/soak/edge_fused_soak.py
cat $L
