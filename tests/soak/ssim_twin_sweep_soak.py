#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: the SSIM / MS-SSIM kernels against the float64 scipy twin (oracle/twin_ssim.py, the
SECOND statement of the two metrics) on random sizes -- odd sides, sides around the strip / segment borders of k_ssim_stream and around
the 176-pixel limit of five scales --, NV12 and RGB8 input: per-scale window means 5e-6 relative, scores 1e-6.
usage: ssim_twin_sweep_soak.py [cases] [max_side]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa: F401
from tm_pkg import tm
from oracle import oracle as O
from oracle import twin_ssim as T
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
max_side = int(sys.argv[2]) if len(sys.argv) > 2 else 700
tm.init_hip(0); tm.set_placement_candidates(1)
rng = np.random.default_rng(31)
edges = [11, 12, 21, 117, 118, 119, 128, 129, 175, 176, 177, 191, 192, 193, 236, 237, 351, 352, 353]
t0, bad, worst_mean, worst_score = time.time(), 0, 0.0, 0.0
for case in range(cases):
    w = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(11, max_side))
    h = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(11, max_side))
    ms = w >= 176 and h >= 176
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssim=True, msssim=ms), batch=1)
    eng.set_full_sums(True)
    if rng.random() < 0.5:
        (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, int(rng.integers(0, 1000)))
        got = eng.compute_one(tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
        lr, ld = O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, 8, 0), O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, 8, 0)
    else:
        r8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        r8[: h // 2] = r8[: h // 2] // 4 + 96
        d8 = np.clip(r8.astype(np.int32) + rng.integers(-9, 10, r8.shape), 0, 255).astype(np.uint8)
        got = eng.compute_one(tm.HwFrame.rgb(r8), tm.HwFrame.rgb(d8))
        lr, ld = O.rgb8_to_linear(r8), O.rgb8_to_linear(d8)
    nsc = 5 if ms else 1
    want = T.scale_means(lr, ld, nsc, odd="drop")
    counts, sw, sh = [], w, h
    for _ in range(nsc):
        counts.append((sw - 10) * (sh - 10)); sw //= 2; sh //= 2
    means = eng.ssim_sums(0)[:, :nsc] / np.asarray(counts, np.float64)[None, :, None]
    rel = float(np.max(np.abs(means / want - 1)))
    ds_ = abs(got.ssim - T.ssim(lr, ld))
    dm = abs(got.msssim - T.msssim(lr, ld)) if ms else 0.0
    worst_mean, worst_score = max(worst_mean, rel), max(worst_score, ds_, dm)
    # a single window (or a handful) over a flat patch is where the f32 statement loses digits: tests/test_ssim_twin.py; bound 5e-5 there
    tol = 5e-6 if min(counts) >= 64 else 5e-5
    if not (rel <= tol and ds_ <= 1e-6 and dm <= 1e-6):
        bad += 1
        print(f"MISMATCH case {case}: {w}x{h} means rel {rel:.2e} ssim {ds_:.2e} msssim {dm:.2e}", flush=True)
    eng.close()
print(f"ssim twin sweep: {cases} random sizes against the float64 twin, mismatches {bad}, worst mean rel {worst_mean:.2e}, worst score diff {worst_score:.2e}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
