#!/usr/bin/env python3
"""GPU box: does the column pass slow down under sustained load (clock / power management) or is its time fixed per allocation?
Runs the headline batch for a few seconds and prints the stage times per window of 20 steps."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm
w, h, B = 1920, 1080, 64
tm.init_hip(0)
pairs = []
for n in range(8):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
for slot in range(B):
    (rt, rp, rch), (dt, dp, dch) = pairs[slot % 8]
    eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
eng.set_profiling(True)
t0 = time.perf_counter()
for win in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    eng.stage_ms(reset=True)
    for _ in range(20):
        eng.compute_async(); eng.sync()
    ms, n = eng.stage_ms(reset=True)
    print(json.dumps({"t_s": round(time.perf_counter() - t0, 2), "ingest": round(ms[0] / n, 3), "colpass": round(ms[1] / n, 3), "rowpass": round(ms[2] / n, 3)}), flush=True)
