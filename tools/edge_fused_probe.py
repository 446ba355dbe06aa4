#!/usr/bin/env python3
"""GPU box: the EDGE jobs in one kernel (k_blur_edge_fused) against the two-pass kernels -- the 108 sums bit for bit, stage times
(ingest, column pass, row pass, ssim, fused EDGE kernel) and wall time per step.
usage: edge_fused_probe.py [WxH:batch,...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from tm_pkg import tm
tm.init_hip(0)
cases = sys.argv[1] if len(sys.argv) > 1 else "333x203:3,1920x1080:8,1920x1080:64"
for case in cases.split(","):
    size, B = case.split(":"); w, h = (int(v) for v in size.split("x")); B = int(B)
    p016 = w >= 3000
    pairs = []
    for n in range(min(B, 4)):
        if p016:
            (rs, rp, rch), (ds, dp, dch) = tm.synth.p016_pair(w, h, n)
        else:
            (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
        pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
    mk = tm.HwFrame.p016 if p016 else tm.HwFrame.nv12
    for slot in range(B):
        (rt, rp, rch), (dt, dp, dch) = pairs[slot % len(pairs)]
        eng.set_pair(slot, mk(rt, rp, rch), mk(dt, dp, dch))
    row = {"case": case}
    ref = None
    for name, variant in (("two_pass", tm.ffi.TM_VARIANT_TWO_PASS_EDGE), ("fused", tm.ffi.TM_VARIANT_FUSED_EDGE)):
        eng.set_variant(variant)
        eng.set_profiling(False)
        for _ in range(5):
            eng.compute_async(); eng.sync()
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            eng.compute_async(); eng.sync()
        row[name + "_wall_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        eng.set_profiling(True)
        eng.stage_ms(reset=True)
        for _ in range(30):
            eng.compute_async(); eng.sync()
        ms, n = eng.stage_ms(reset=True)
        row[name + "_stage_ms[ingest,col,row,ssim,edge]"] = [round(m / n, 4) for m in ms]
        got = np.stack([eng.raw_sums(i) for i in range(B)])
        if ref is None:
            ref = got
        else:
            row["sums_bit_identical"] = bool(np.array_equal(ref.view(np.uint64), got.view(np.uint64)))
            row["max_rel_diff"] = float(np.max(np.abs(ref - got) / np.maximum(np.abs(ref), 1e-300)))
    print(json.dumps(row), flush=True)
    eng.close()
