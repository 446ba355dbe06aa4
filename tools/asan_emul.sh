#!/bin/bash
# AddressSanitizer run of the kernel source through the CPU emulator (tests/emul): out-of-bounds accesses of any kernel
# against arenas of exactly the engine's size.  usage: tools/asan_emul.sh [pytest -k expression]
cd "$(dirname "$0")/.."
export TM_EMUL_ASAN=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=0
LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/test_emul_kernels.py -x -q -k "${1:-wave_ingest or p016_launch or ssim}" 2>&1 | tail -25
