#!/usr/bin/env python3
"""GPU box: does the column-pass time depend on the engine's allocation?  Creates the engine several times in one process
(optionally keeping the previous ones alive so that the arenas land elsewhere) and prints the stage times of each."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--engines", type=int, default=5); ap.add_argument("--keep", type=int, default=0)
a = ap.parse_args()
w, h = 1920, 1080
tm.init_hip(0)
pairs = []
for n in range(4):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
kept = []
for i in range(a.engines):
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=a.batch)
    for slot in range(a.batch):
        (rt, rp, rch), (dt, dp, dch) = pairs[slot % 4]
        eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
    eng.set_profiling(True)
    for _ in range(30):
        eng.compute_async(); eng.sync()
    eng.stage_ms(reset=True)
    for _ in range(30):
        eng.compute_async(); eng.sync()
    ms, n = eng.stage_ms(reset=True)
    print(json.dumps({"engine": i, "ingest_ms": round(ms[0] / n, 3), "blur_v_ms": round(ms[1] / n, 3), "blur_h_ms": round(ms[2] / n, 3)}), flush=True)
    if i < a.keep:
        kept.append(eng)
    else:
        eng.close()
