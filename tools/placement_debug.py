#!/usr/bin/env python3
"""GPU box: the placement search with its candidates' addresses and times printed (tm_set_debug_log), a few engines in a row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm
tm.init_hip(0)
tm.set_debug_log(True)
tm.set_placement_candidates(8)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng = tm.TurboMetrics(1920, 1080, tm.Metrics(ssimulacra2=True), batch=64)
    print("engine", i, flush=True)
    eng.close()
