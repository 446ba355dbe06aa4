#!/usr/bin/env python3
"""GPU box: the placement search with its candidates' addresses and times printed (TM_PLACEMENT_DEBUG), a few engines in a row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["TM_PLACEMENT_DEBUG"] = "1"
import torch
from tm_pkg import tm
tm.init_hip(0)
tm.set_placement_candidates(8)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng = tm.TurboMetrics(1920, 1080, tm.Metrics(ssimulacra2=True), batch=64)
    print("engine", i, flush=True)
    eng.close()
