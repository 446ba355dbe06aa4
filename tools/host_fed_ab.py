#!/usr/bin/env python3
"""GPU box: page-locked host frames -> two engines taking turns (bench.py's host_fed leg), A/B on one box, interleaved:
upload streams 1 / 2 (TM_DBG_UPLOAD_STREAMS) x frame layout (NV12 / P016 surface: one 2-D copy; planar picture as in a Y4M file: one
linear copy, or 2-D copies with TM_DBG_LINEAR_UPLOAD = 0).  usage: host_fed_ab.py [1080p|4k] [rounds] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
F = tm.ffi
size = sys.argv[1] if len(sys.argv) > 1 else "1080p"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h, bits = (1920, 1080, 8) if size == "1080p" else (3840, 2160, 10)
B = int(sys.argv[3]) if len(sys.argv) > 3 else (16 if size == "1080p" else 4)
steps = 80 if size == "1080p" else 60
tm.init_hip(0); tm.set_placement_candidates(1)
dt = np.uint8 if bits == 8 else np.uint16
cw, ch = (w + 1) // 2, (h + 1) // 2
surf, planar = [], []
for n in range(4):
    pair = tm.synth.yuv420_pair(w, h, n, bits)
    s2, p2 = [], []
    for planes in pair:
        sf, pitch, coded = tm.synth.pack_biplanar(planes, w, h, bits)
        s2.append((torch.from_numpy(sf).pin_memory(), pitch, coded))
        flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(p.astype(dt)).reshape(-1) for p in planes])).pin_memory()
        p2.append((flat[:w * h].view(h, w), flat[w * h:w * h + cw * ch].view(ch, cw), flat[w * h + cw * ch:].view(ch, cw)))
    surf.append(s2); planar.append(p2)
mk = tm.HwFrame.nv12 if bits == 8 else tm.HwFrame.p016
in_bytes = w * h * 3 // 2 * (1 if bits == 8 else 2) * 2


def run(layout, streams, linear):
    engs = [tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B) for _ in range(2)]
    for e in engs:
        e.debug_set_param(F.TM_DBG_UPLOAD_STREAMS, streams); e.debug_set_param(F.TM_DBG_LINEAR_UPLOAD, linear)
    busy = [False, False]

    def submit(k):
        e = engs[k & 1]
        if busy[k & 1]:
            e.sync(); e.scores_batch(B)
        for slot in range(B):
            n = (k * B + slot) % 4
            if layout == "surface":
                e.set_pair(slot, mk(*surf[n][0]), mk(*surf[n][1]))
            else:
                e.set_pair(slot, tm.HwFrame.i420(*planar[n][0], bits=bits), tm.HwFrame.i420(*planar[n][1], bits=bits))
        e.compute_async(B); busy[k & 1] = True
    for k in range(6):
        submit(k)
    for e in engs:
        e.sync()
    busy[:] = [False, False]
    t0 = time.perf_counter()
    for k in range(steps):
        submit(k)
    for e in engs:
        e.sync()
    dtm = time.perf_counter() - t0
    for e in engs:
        e.close()
    return B * steps / dtm, B * steps * in_bytes / dtm / 1e9


for r in range(rounds):
    for layout, streams, linear in (("surface", 1, 1), ("surface", 2, 1), ("planar", 1, 0), ("planar", 2, 0), ("planar", 1, 1), ("planar", 2, 1)):
        v, gbs = run(layout, streams, linear)
        print(f"{size} x{B}  {layout:8s} upload streams {streams}  {'linear' if linear and layout == 'planar' else '2-D   '}: {v:7.0f} pairs/s  {gbs:5.1f} GB/s", flush=True)
