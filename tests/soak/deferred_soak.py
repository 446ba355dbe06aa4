#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: compute_one_deferred / collect under repetition -- tens of thousands of one-pair submissions with two in flight (two
engines taking turns), frames from device memory, page-locked and pageable host memory in turn, collected with a lag of one and,
every so often, out of order or after a blocking compute_one in between; every score must be compute_one's, bit for bit.
usage: deferred_soak.py [submissions]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
tm.init_hip(0)
t_start = time.time()
for w, h, share in ((640, 360, 0.5), (1920, 1080, 0.5)):
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=1)
    kinds, want = [], []
    for n in range(6):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
        want.append(eng.compute_one(tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch)))
        dev = (tm.HwFrame.nv12(torch.from_numpy(rs).cuda(), rp, rch), tm.HwFrame.nv12(torch.from_numpy(ds).cuda(), dp, dch))
        pin = (tm.HwFrame.nv12(torch.from_numpy(np.asarray(rs).copy()).pin_memory(), rp, rch), tm.HwFrame.nv12(torch.from_numpy(np.asarray(ds).copy()).pin_memory(), dp, dch))
        pag = (tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
        kinds.append((dev, pin, pag))
    torch.cuda.synchronize()
    steps = int(N * share)
    bad, last, t0 = 0, None, time.time()
    rng = np.random.default_rng(1)
    for k in range(steps):
        n, m = int(rng.integers(6)), (0 if k % 7 else int(rng.integers(3)))  # mostly device frames; now and then pinned / pageable
        t = eng.compute_one_deferred(*kinds[n][m])
        if last is not None:
            if k % 1000 == 999:  # a blocking call in between: the pairs in flight are finished first and stay collectable
                bad += eng.compute_one(*kinds[(n + 1) % 6][0]) != want[(n + 1) % 6]
            bad += eng.collect(last[0]) != want[last[1]]
        last = (t, n)
    bad += eng.collect(last[0]) != want[last[1]]
    # round 6: three to eight pairs in flight (set_deferred_depth), the depth changed every few hundred submissions with pairs in flight
    tickets, depth = [], 2
    for k in range(steps // 2):
        if k % 300 == 0:
            depth = int(rng.integers(2, 9))
            eng.set_deferred_depth(depth)
        n, m = int(rng.integers(6)), (0 if k % 7 else int(rng.integers(3)))
        tickets.append((eng.compute_one_deferred(*kinds[n][m]), n))
        while len(tickets) >= depth + (3 if k % 300 < 3 else 0):  # (right after a change: older tickets stay uncollected for a few calls)
            t, i = tickets.pop(0)
            bad += eng.collect(t) != want[i]
    for t, i in tickets:
        bad += eng.collect(t) != want[i]
    dt = time.time() - t0
    print(f"{w}x{h}: {steps} deferred submissions, {steps / dt:.0f} pairs/s, mismatches {bad}", flush=True)
    assert bad == 0
    eng.close()
print(f"deferred soak ok in {time.time() - t_start:.0f} s")
