#!/usr/bin/env python3
"""GPU box: same-process A/B of the ingest stage.  One engine (so the blur passes see one arena), the ingest kernel switched per
window of 20 steps between the tile kernel (TM_VARIANT_TILE_INGEST) and the row-walking kernel at several rows_per_wave.

    python tools/ingest_ab.py [1080p_nv12|4k_p016] [--metrics psnr,msssim,ssimulacra2] [--batch N] [--rows 2,4,8,16,32]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm

args = sys.argv[1:]
wl = args[0] if args and not args[0].startswith("--") else "1080p_nv12"
w, h, B, gen, mk = (1920, 1080, 64, tm.synth.nv12_pair, tm.HwFrame.nv12) if wl == "1080p_nv12" else (3840, 2160, 24, tm.synth.p016_pair, tm.HwFrame.p016)
if "--batch" in args: B = int(args[args.index("--batch") + 1])
mets = args[args.index("--metrics") + 1].split(",") if "--metrics" in args else ["ssimulacra2"]
rows = [int(x) for x in (args[args.index("--rows") + 1].split(",") if "--rows" in args else "2,4,8,16,32".split(","))]
tm.init_hip(0)
tm.set_placement_candidates(1)
pairs = []
ND = int(args[args.index("--distinct") + 1]) if "--distinct" in args else 4
for n in range(ND):
    (rs, rp, rch), (ds, dp, dch) = gen(w, h, n)
    pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
L = tm.ffi.lib()
eng = tm.TurboMetrics(w, h, tm.Metrics(**{m: True for m in mets}), batch=B)
for slot in range(B):
    (rt, rp, rch), (dt, dp, dch) = pairs[slot % ND]
    eng.set_pair(slot, mk(rt, rp, rch), mk(dt, dp, dch))
eng.set_profiling(True)
for _ in range(30):
    eng.compute_async(); eng.sync()
ref = None
out = {}
for rnd in range(3):
    for name, variant, r in ([] if "--no-tile" in args else [("tile", tm.ffi.TM_VARIANT_TILE_INGEST, 0)]) + [(f"rows{r}", 0, r) for r in rows]:
        eng.set_variant(variant)
        L.tm_engine_debug_set_ingest_rows(eng._h, r)
        eng.compute_async(); eng.sync()
        eng.stage_ms(reset=True)
        for _ in range(20):
            eng.compute_async(); eng.sync()
        ms, n = eng.stage_ms(reset=True)
        sc = [eng.scores(i).ssimulacra2 for i in range(min(B, 4))] if "ssimulacra2" in mets else [eng.sse(0)]
        if ref is None: ref = sc
        assert sc == ref, (name, sc, ref)
        out.setdefault(name, []).append([round(m / n, 4) for m in ms] + [round(sum(ms) / n, 4)])
print(json.dumps({"workload": wl, "batch": B, "metrics": mets, "stage_ms_per_step[ingest,col,row,ssim,sum]": out, "scores_identical": True}))
