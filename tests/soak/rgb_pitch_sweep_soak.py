#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  GPU box: packed RGB frames (u8, u16, f32, already-linear f32) handed over through the C ABI with random row
pitches, from device / page-locked / pageable memory, against the same pixels handed over tight: raw sums and SSE bit for bit.
usage: rgb_pitch_sweep_soak.py [cases]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
F = tm.ffi
L = F.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tm.init_hip(0); tm.set_placement_candidates(1)
rng = np.random.default_rng(8)
t0, bad = time.time(), 0
FN = {"rgb8": (L.tm_engine_set_frame_rgb8, np.uint8), "rgb16": (L.tm_engine_set_frame_rgb16, np.uint16), "rgbf32": (L.tm_engine_set_frame_rgbf32, np.float32),
      "linear": (L.tm_engine_set_frame_linear_f32, np.float32)}
for case in range(cases):
    w, h = int(rng.integers(1, 500)), int(rng.integers(1, 400))
    kind = str(rng.choice(list(FN)))
    fn, dt = FN[kind]
    B = int(rng.integers(1, 4))
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B)
    base = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B)
    keep = []
    for slot in range(B):
        r8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        d8 = np.clip(r8.astype(np.int32) + rng.integers(-12, 13, r8.shape), 0, 255).astype(np.uint8)
        for side, img in enumerate((r8, d8)):
            if dt == np.uint8: a = img
            elif dt == np.uint16: a = (img.astype(np.uint16) * 257 + int(rng.integers(0, 200))).astype(np.uint16)
            else: a = (img.astype(np.float32) / 255.0).astype(np.float32)
            a = np.ascontiguousarray(a)
            keep.append(a)
            row = w * 3 * a.itemsize
            assert fn(base._h, slot, side, a.ctypes.data_as(C.c_void_p), row, F.TM_MEM_HOST) == 0
            pad = int(rng.integers(0, 64)) * a.itemsize
            q = np.zeros((h, row + pad), np.uint8)
            q[:, :row] = a.view(np.uint8).reshape(h, row)
            mem = str(rng.choice(["host", "pinned", "device"]))
            if mem == "host":
                buf = q; ptr = buf.ctypes.data; m = F.TM_MEM_HOST
            elif mem == "pinned":
                buf = torch.from_numpy(q).pin_memory(); ptr = buf.data_ptr(); m = F.TM_MEM_HOST_PINNED
            else:
                buf = torch.from_numpy(q).cuda(); ptr = buf.data_ptr(); m = F.TM_MEM_DEVICE
            keep.append(buf)
            rc = fn(eng._h, slot, side, C.c_void_p(ptr), row + pad, m)
            assert rc == 0, (rc, kind, mem, w, h, row, pad)
    torch.cuda.synchronize()
    base.compute_async(B); base.sync()
    eng.compute_async(B); eng.sync()
    for i in range(B):
        if not (np.array_equal(eng.raw_sums(i), base.raw_sums(i)) and eng.sse(i) == base.sse(i)):
            bad += 1
            print(f"MISMATCH case {case}: {w}x{h} {kind} batch {B} slot {i}", flush=True)
    eng.close(); base.close()
print(f"rgb pitch sweep: {cases} random cases, mismatches {bad}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
