#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: tests/test_gpu_parity.py::test_random_sweep_of_sizes_kinds_and_metric_masks far beyond its ten committed seeds -- random
frame sizes (around the kernels' tile / strip / segment borders), input kinds, colour matrices, metric masks, pruned or full sums,
batches: every case against the oracle.  usage: random_sweep_soak.py [first_seed] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch  # noqa: F401  (torch's HIP runtime first, like tests/conftest.py)
from tests import test_gpu_parity as T
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t0, bad = time.time(), []
for seed in range(first, first + count):
    try:
        T.test_random_sweep_of_sizes_kinds_and_metric_masks(seed)
    except Exception:  # noqa: BLE001
        bad.append(seed)
        print(f"seed {seed} FAILED\n{traceback.format_exc()[-1500:]}", flush=True)
print(f"random sweep: seeds {first} .. {first + count - 1}: {count - len(bad)} passed, failed: {bad}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
