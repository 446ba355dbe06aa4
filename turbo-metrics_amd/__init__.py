"""turbo-metrics_amd: MI355X-native SSIMULACRA2 / PSNR frame-pair engine (gfx950 HIP kernels behind a
C ABI) with a host-side mirror of the reference's operator interface.

  ffi     -- ctypes binding of include/turbo_metrics_hip.h (libturbometrics_hip.so, built in-tree)
  engine  -- TurboMetrics / Ssimulacra2 / FrameScores mirrors of the reference types
  synth   -- seeded synthetic frame generators used by bench.py and the tests

There is no CPU implementation in this package: without the HIP library (or without a gfx950 GPU)
the operators raise.
"""
from . import ffi, launch, shard, synth  # noqa: F401
from .engine import (ColorMatrix, FrameScores, HwFrame, Metrics, Ssimulacra2, TmError,  # noqa: F401
                     TurboMetrics, init_hip, set_debug_log, set_placement_candidates)
