#!/bin/bash
# GPU box: the soak programs tools/gpu_soaks.sh does not run (threads, SSIM twin, RGB pitches, create / destroy) + further seeds of the random sweep
set -u
TAG=${1:-r06v}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_more_soaks.log
: > $L
run() { timeout "$1" python "${@:2}" 2>&1 | grep -v amdgpu.ids | tail -3 >> $L; }
run 600 tests/soak/thread_soak.py 8 200
run 600 tests/soak/ssim_twin_sweep_soak.py 300 700
run 600 tests/soak/rgb_pitch_sweep_soak.py 1500
run 400 tests/soak/create_destroy_soak.py 200
run 600 tests/soak/random_sweep_soak.py 40000 8000
run 900 tests/soak/surface_sweep_soak.py 6000
cat $L
