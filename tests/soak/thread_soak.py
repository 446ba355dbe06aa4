#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: engines created, used and destroyed from several host threads at once (each engine by one thread at a time, as the header
asks; the library's process-wide pieces -- the device's shared side stream and second upload stream, their reference counts, the
registry of live engines -- are what is exercised).  Every result is compared with the expected one.  usage: thread_soak.py [threads] [cycles]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CYC = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tm.init_hip(0); tm.set_placement_candidates(1)
SIZES = [(160, 96), (640, 360), (333, 203), (1920, 1080)]
exp = {}
for (w, h) in SIZES:
    one = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=1)
    fr = []
    for n in range(3):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
        s = one.compute_one(tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
        fr.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch), (rs, ds), s, one.raw_sums(0).copy()))
    exp[(w, h)] = fr
    one.close()
torch.cuda.synchronize()
errors = []


def worker(tid):
    try:
        rng = np.random.default_rng(100 + tid)
        tm.init_hip(0)
        for c in range(CYC):
            w, h = SIZES[int(rng.integers(0, len(SIZES)))]
            B = int(rng.integers(1, 7 if w < 1000 else 9))
            eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B)
            if rng.random() < 0.5:
                eng.set_variant(tm.ffi.TM_VARIANT_FUSED_EDGE)  # the side stream, also for small launches
            picks = [int(rng.integers(0, 3)) for _ in range(B)]
            for rep in range(int(rng.integers(1, 4))):
                for slot, p in enumerate(picks):
                    (rt, rp, rch), (dt, dp, dch), (rs, ds), _, _ = exp[(w, h)][p]
                    if rng.random() < 0.3:  # pageable host frames: uploads on the engine's stream
                        eng.set_pair(slot, tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
                    else:
                        eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
                eng.compute_async(B); eng.sync()
                for slot, p in enumerate(picks):
                    want = exp[(w, h)][p]
                    m = want[4] != 0
                    if not (np.array_equal(eng.raw_sums(slot)[m], want[4][m]) and eng.scores(slot) == want[3]):
                        errors.append((tid, c, w, h, B, slot))
            eng.close()
    except Exception as ex:  # noqa: BLE001
        errors.append((tid, repr(ex)))


t0 = time.time()
th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
for t in th: t.start()
for t in th: t.join()
print(f"thread soak: {T} threads x {CYC} engine lifetimes, errors {len(errors)} {errors[:5]}, {time.time() - t0:.0f} s")
sys.exit(1 if errors else 0)
