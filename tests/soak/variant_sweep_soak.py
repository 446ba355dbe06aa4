#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: kernel variants against each other on random sizes far beyond the committed cases -- for each (w, h, batch, kind) the raw sums
and SSE of the default configuration, of the fused EDGE kernel forced, of the two passes forced, of the eight-wave and the one-wave
row pass, of the tile ingest kernel and of the straight-line reference pipeline must be identical bit for bit.
usage: variant_sweep_soak.py [cases] [max_side]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
F = tm.ffi
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
max_side = int(sys.argv[2]) if len(sys.argv) > 2 else 1400
tm.init_hip(0)
tm.set_placement_candidates(1)
rng = np.random.default_rng(20261003)
edges = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025]
VARIANTS = [("fused", F.TM_VARIANT_FUSED_EDGE), ("two_pass", F.TM_VARIANT_TWO_PASS_EDGE), ("split_rows", F.TM_VARIANT_SPLIT_ROWS | F.TM_VARIANT_FUSED_EDGE),
            ("whole_rows", F.TM_VARIANT_WHOLE_ROWS | F.TM_VARIANT_TWO_PASS_EDGE), ("tile_ingest", F.TM_VARIANT_TILE_INGEST), ("reference", F.TM_VARIANT_REFERENCE),
            ("upper_kernel", F.TM_VARIANT_UPPER_KERNEL)]  # round 6: pyramid levels 2..5 by k_ingest_upper_rd instead of the ingest kernel's own epilogue
t0, bad = time.time(), 0
for case in range(cases):
    w = int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(1, max_side))
    h = int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(1, max_side))
    B = int(rng.integers(1, 5))
    r = rng.random()
    p016, p10 = r < 0.3, r >= 0.8  # (p10, round 6: 10-bit planes packed three samples to a word)
    gen, mk = (tm.synth.p016_pair, tm.HwFrame.p016) if p016 else (tm.synth.nv12_pair, tm.HwFrame.nv12)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B)
    keep = []
    for slot in range(B):
        if p10:
            ref, dis = tm.synth.yuv420_pair(w, h, int(rng.integers(0, 1000)), 10)
            fr = [tm.HwFrame.i420p10(*(torch.from_numpy(tm.synth.p10_pack_plane(p).view(np.int32)).cuda() for p in side)) for side in (ref, dis)]
            keep.append(fr)
            eng.set_pair(slot, fr[0], fr[1])
            continue
        (rs, rp, rch), (ds, dp, dch) = gen(w, h, int(rng.integers(0, 1000)))
        eng.set_pair(slot, mk(torch.from_numpy(rs).cuda(), rp, rch), mk(torch.from_numpy(ds).cuda(), dp, dch))
    eng.compute_async(B); eng.sync()
    want = [(eng.raw_sums(i).copy(), eng.sse(i)) for i in range(B)]
    for name, v in VARIANTS:
        eng.set_variant(v)
        eng.compute_async(B); eng.sync()
        for i in range(B):
            if not (np.array_equal(eng.raw_sums(i), want[i][0]) and eng.sse(i) == want[i][1]):
                bad += 1
                print(f"MISMATCH case {case}: {w}x{h} batch {B} {'p016' if p016 else 'p10' if p10 else 'nv12'} variant {name} slot {i}", flush=True)
    eng.close()
print(f"variant sweep: {cases} random cases x {len(VARIANTS)} variants against the default configuration, mismatches {bad}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
