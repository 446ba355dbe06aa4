#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: ONE long-lived engine under a random sequence of state changes -- kernel variant, hipGraph replay on / off, pruned / full
sums, channel mode, profiling, slots per launch, frames of other KINDS (NV12, P016, planar 8 / 10 bit, packed 10 bit, RGB8 / 16 / f32) and memory kinds
in random slots -- every launch checked against what a fresh one-pair engine computes for the same frames (raw sums where a weight
reads them, SSE, SSIM sums).  State that leaks from one launch into the next shows up here.  usage: engine_state_soak.py [launches] [w h]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
F = tm.ffi
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (416, 240)
tm.init_hip(0); tm.set_placement_candidates(1)
rng = np.random.default_rng(4242)
B = 6
metrics = tm.Metrics(ssimulacra2=True, psnr=True, ssim=True, msssim=(w >= 176 and h >= 176))
eng = tm.TurboMetrics(w, h, metrics, batch=B)
one = tm.TurboMetrics(w, h, metrics, batch=1)
one.set_full_sums(True)

# a pool of frame pairs of every kind, each with its expected results from the one-pair engine
pool = []
for n in range(16):
    kind = ["nv12", "p016", "i420_8", "i420_10", "rgb8", "rgb16", "rgbf32", "i420_p10"][n % 8]
    if kind in ("nv12", "p016"):
        (rs, rp, rch), (ds, dp, dch) = (tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair)(w, h, n)
        mk = tm.HwFrame.nv12 if kind == "nv12" else tm.HwFrame.p016
        host = (mk(rs, rp, rch), mk(ds, dp, dch))
        dev = (mk(torch.from_numpy(rs).cuda(), rp, rch), mk(torch.from_numpy(ds).cuda(), dp, dch))
        pin = (mk(torch.from_numpy(np.asarray(rs).copy()).pin_memory(), rp, rch), mk(torch.from_numpy(np.asarray(ds).copy()).pin_memory(), dp, dch))
    elif kind == "i420_p10":  # 10-bit planes packed three samples to a word (round 6)
        pr = tm.synth.yuv420_pair(w, h, n, 10)
        mkp = lambda planes, f: tm.HwFrame.i420p10(*[f(tm.synth.p10_pack_plane(p).view(np.int32)) for p in planes])
        host = (mkp(pr[0], lambda a: a.view(np.uint32)), mkp(pr[1], lambda a: a.view(np.uint32)))
        dev = (mkp(pr[0], lambda a: torch.from_numpy(a).cuda()), mkp(pr[1], lambda a: torch.from_numpy(a).cuda()))
        pin = (mkp(pr[0], lambda a: torch.from_numpy(a).pin_memory()), mkp(pr[1], lambda a: torch.from_numpy(a).pin_memory()))
    elif kind.startswith("i420"):
        bits = 8 if kind == "i420_8" else 10
        pr = tm.synth.yuv420_pair(w, h, n, bits)
        dt = np.uint8 if bits == 8 else np.uint16
        mkp = lambda planes, f: tm.HwFrame.i420(*[f(np.ascontiguousarray(p.astype(dt))) for p in planes], bits=bits)
        host = (mkp(pr[0], lambda a: a), mkp(pr[1], lambda a: a))
        dev = (mkp(pr[0], lambda a: torch.from_numpy(a).cuda()), mkp(pr[1], lambda a: torch.from_numpy(a).cuda()))
        pin = (mkp(pr[0], lambda a: torch.from_numpy(a).pin_memory()), mkp(pr[1], lambda a: torch.from_numpy(a).pin_memory()))
    else:
        r8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        d8 = np.clip(r8.astype(np.int32) + rng.integers(-9, 10, r8.shape), 0, 255).astype(np.uint8)
        if kind == "rgb8": a, b = r8, d8
        elif kind == "rgb16": a, b = r8.astype(np.uint16) * 257, (d8.astype(np.uint16) * 257 + 31).astype(np.uint16)
        else: a, b = r8.astype(np.float32) / 255, d8.astype(np.float32) / 255
        host = (tm.HwFrame.rgb(a), tm.HwFrame.rgb(b))
        dev = (tm.HwFrame.rgb(torch.from_numpy(a).cuda()), tm.HwFrame.rgb(torch.from_numpy(b).cuda()))
        pin = (tm.HwFrame.rgb(torch.from_numpy(a.copy()).pin_memory()), tm.HwFrame.rgb(torch.from_numpy(b.copy()).pin_memory()))
    one.compute_one(*host)
    pool.append({"kind": kind, "frames": (dev, pin, host), "sums": one.raw_sums(0).copy(), "sse": one.sse(0), "ssums": one.ssim_sums(0).copy(), "score": one.scores(0)})
torch.cuda.synchronize()
from oracle import oracle as O  # weights only (test infrastructure; this tool is one)
wmask = (O.weights().reshape(3, 6, 6) != 0.0).transpose(1, 2, 0)

slots = [None] * B
full = False
state = {"variant": 0, "graph": None, "full": False, "first": False, "prof": False}
VARS = [0, F.TM_VARIANT_FUSED_EDGE, F.TM_VARIANT_TWO_PASS_EDGE, F.TM_VARIANT_SPLIT_ROWS, F.TM_VARIANT_WHOLE_ROWS, F.TM_VARIANT_TILE_INGEST,
        F.TM_VARIANT_TILE_INGEST | F.TM_VARIANT_FUSED_EDGE, F.TM_VARIANT_WIDE_ROWS, F.TM_VARIANT_UPPER_KERNEL, F.TM_VARIANT_UPPER_KERNEL | F.TM_VARIANT_FUSED_EDGE]
t0, bad = time.time(), 0
hold, n = 0, 1
for k in range(launches):
    # runs of launches of one shape with no state change in between (round 6: from the fourth one on such launches of up to four pairs
    # replay a captured graph in the default mode -- with other frames in the slots, and captured again after every change)
    if hold == 0 and rng.random() < 0.1:
        hold = int(rng.integers(4, 14))
    for _ in range(0 if hold else int(rng.integers(0, 4))):  # a few random state changes
        op = int(rng.integers(0, 6))
        if op == 0: state["variant"] = int(rng.choice(VARS)); eng.set_variant(state["variant"])
        elif op == 1: state["graph"] = [None, None, True, False][int(rng.integers(0, 4))]; eng.set_graph(state["graph"])  # None: the default (repeating small launches replay)
        elif op == 2: state["full"] = bool(rng.integers(0, 2)); eng.set_full_sums(state["full"])
        elif op == 3: state["first"] = bool(rng.integers(0, 2)); eng.set_channel_mode(state["first"])
        elif op == 4: state["prof"] = bool(rng.integers(0, 2)); eng.set_profiling(state["prof"])
        else: eng.debug_set_param(F.TM_DBG_UPLOAD_STREAMS, int(rng.integers(1, 3)))
    if hold:
        hold -= 1
    else:
        n = int(rng.integers(1, B + 1))
    for slot in range(B):  # some slots get other frames (always the first n if never set)
        if slots[slot] is None or rng.random() < 0.4:
            p = int(rng.integers(0, len(pool)))
            if rng.random() < 0.25:  # ref and dis of a slot from different memory kinds
                eng.set_frame(slot, 0, pool[p]["frames"][int(rng.integers(0, 3))][0]); eng.set_frame(slot, 1, pool[p]["frames"][int(rng.integers(0, 3))][1])
            else:
                eng.set_pair(slot, *pool[p]["frames"][int(rng.integers(0, 3))])
            slots[slot] = p
    eng.compute_async(n); eng.sync()
    for slot in range(n):
        e = pool[slots[slot]]
        got = eng.raw_sums(slot)
        m = np.ones_like(wmask) if state["full"] else wmask
        ok = np.array_equal(got[m], e["sums"][m]) and eng.sse(slot) == e["sse"]
        sc = eng.scores(slot)
        ok = ok and sc.ssimulacra2 == e["score"].ssimulacra2
        if not state["first"]:
            ok = ok and sc.psnr == e["score"].psnr and sc.ssim == e["score"].ssim and sc.msssim == e["score"].msssim
        if not ok:
            bad += 1
            print(f"MISMATCH launch {k} slot {slot} ({e['kind']}), n {n}, state {state}", flush=True)
print(f"engine state soak: {launches} launches of up to {B} slots at {w}x{h} with random state changes in between, mismatches {bad}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
