#!/usr/bin/env python3
"""GPU A/B: per-stage HIP-event times for each kernel variant, interleaved rounds in one process."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="1080p"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--variants", default="1033,777"); ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--full-sums", action="store_true")
a = ap.parse_args()
w, h, gen, mk = (1920, 1080, tm.synth.nv12_pair, tm.HwFrame.nv12) if a.workload == "1080p" else (3840, 2160, tm.synth.p016_pair, tm.HwFrame.p016)
tm.init_hip(0)
eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=a.batch)
eng.set_full_sums(a.full_sums)
keep = []
for n in range(4):
    (rs, rp, rch), (ds, dp, dch) = gen(w, h, n)
    keep.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
for slot in range(a.batch):
    (rt, rp, rch), (dt, dp, dch) = keep[slot % 4]
    eng.set_pair(slot, mk(rt, rp, rch), mk(dt, dp, dch))
torch.cuda.synchronize()
eng.set_profiling(True)
res = {}
for rnd in range(a.rounds):
    for v in [int(x) for x in a.variants.split(",")]:
        eng.set_variant(v)
        eng.compute_async(); eng.sync(); eng.stage_ms(reset=True)
        for _ in range(a.steps):
            eng.compute_async(); eng.sync()
        ms, n = eng.stage_ms(reset=True)
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.compute_async(); eng.sync()
        wall = (time.perf_counter() - t0) / a.steps * 1e3
        res.setdefault(v, []).append([m / max(n, 1) for m in ms] + [wall])
px = sum(((w + (1 << s) - 1) >> s) * ((h + (1 << s) - 1) >> s) for s in range(6))
for v, rows in res.items():
    r = np.array(rows); med = np.median(r, axis=0)
    gb = 84 * px * a.batch / 1e9
    print(json.dumps({"variant": v, "ingest_ms": round(med[0], 3), "blur_v_ms": round(med[1], 3), "blur_h_ms": round(med[2], 3),
                      "blur_v_GBs": round(gb / med[1] * 1e3, 1), "blur_h_GBs": round(gb / med[2] * 1e3, 1),
                      "wall_ms_per_step": round(med[3], 3), "pairs_per_s_wall": round(a.batch / (med[3] * 1e-3), 1)}))
