#!/usr/bin/env python3
"""GPU box: how do the column / row pass react to WHERE the pass-1 arena starts?  For each of a few fresh allocations
(engine creations, placement search off) the arena's start is moved through a set of offsets inside the same allocation
(tm_engine_debug_set_v_offset) and the stage times are printed: if the spread between allocations is reproduced by offsets
inside one allocation, an alignment rule can replace the placement search; if not, it is the physical backing."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--engines", type=int, default=3); ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--offsets", default="0,256,512,1024,2048,4096,4352,8192,16384,32768,65536,131072,262144,524288,1048576,2097152,3145728")
a = ap.parse_args()
w, h = 1920, 1080
tm.init_hip(0)
tm.set_placement_candidates(1)
pairs = []
for n in range(4):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
L = tm.ffi.lib()
kept = []
for i in range(a.engines):
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=a.batch)
    for slot in range(a.batch):
        (rt, rp, rch), (dt, dp, dch) = pairs[slot % 4]
        eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
    eng.set_profiling(True)
    for _ in range(40):
        eng.compute_async(); eng.sync()
    base = eng.scores(0).ssimulacra2
    row = []
    for off in [int(x) for x in a.offsets.split(",")]:
        assert L.tm_engine_debug_set_v_offset(eng._h, off) == 0
        for _ in range(3):
            eng.compute_async(); eng.sync()
        eng.stage_ms(reset=True)
        for _ in range(a.steps):
            eng.compute_async(); eng.sync()
        ms, n = eng.stage_ms(reset=True)
        assert eng.scores(0).ssimulacra2 == base
        row.append((off, round(ms[1] / n, 3), round(ms[2] / n, 3)))
    print(json.dumps({"engine": i, "offset_colpass_rowpass_ms": row}), flush=True)
    kept.append(eng)  # keep it alive: the next engine's arena lands elsewhere
