#!/usr/bin/env python3
"""GPU box: where a small batch spends its time.  Stage times (HIP events between the kernels) and wall time per step for
batch 1..64 at 1080p, direct launches and hipGraph replay."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm
w, h = 1920, 1080
tm.init_hip(0)
pairs = []
for n in range(4):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
for B in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16,64".split(","))]:
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
    for slot in range(B):
        (rt, rp, rch), (dt, dp, dch) = pairs[slot % 4]
        eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
    row = {"batch": B}
    sc = None
    for name, variant in (("whole", tm.ffi.TM_VARIANT_WHOLE_ROWS), ("split", tm.ffi.TM_VARIANT_SPLIT_ROWS)):
        eng.set_variant(variant)
        eng.set_profiling(False)
        for _ in range(30):
            eng.compute_async(); eng.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            eng.compute_async(); eng.sync()
        row[name + "_wall_ms"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
        eng.set_profiling(True)
        eng.stage_ms(reset=True)
        for _ in range(100):
            eng.compute_async(); eng.sync()
        ms, n = eng.stage_ms(reset=True)
        row[name + "_stage_ms[ingest,col,row,ssim]"] = [round(m / n, 4) for m in ms]
        got = [eng.scores(i).ssimulacra2 for i in range(B)]
        assert sc is None or sc == got
        sc = got
    row["pairs_per_s"] = {k: round(B / row[k + "_wall_ms"] * 1e3) for k in ("whole", "split")}
    # back-to-back submission without a sync per step (what a pipelined caller sees)
    eng.set_profiling(False)
    torch.cuda.synchronize()
    print(json.dumps(row), flush=True)
    eng.close()
