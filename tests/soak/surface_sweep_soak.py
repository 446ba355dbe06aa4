#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: every way a 4:2:0 frame can be handed over -- declared surface (random pitch and coded height), planar I420 (random or
tight pitches, with and without the one-copy upload), from device / page-locked / pageable memory, with upload fences or without,
in random slots of a random batch -- against the same samples as tight device surfaces: raw sums and SSE bit for bit.
usage: surface_sweep_soak.py [cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
F = tm.ffi
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tm.init_hip(0); tm.set_placement_candidates(1)
rng = np.random.default_rng(77)
t0, bad = time.time(), 0


def place(arr, mem):
    if mem == "device":
        return torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    if mem == "pinned":
        return torch.from_numpy(np.ascontiguousarray(arr)).pin_memory()
    return np.ascontiguousarray(arr)


for case in range(cases):
    w, h = int(rng.integers(2, 900)), int(rng.integers(2, 700))
    bits = int(rng.choice([8, 10, 12, 16]))
    B = int(rng.integers(1, 5))
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B)
    base = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B)
    eng_lin, eng_str = int(rng.integers(0, 2)), int(rng.integers(1, 3))
    eng.debug_set_param(F.TM_DBG_LINEAR_UPLOAD, eng_lin)
    eng.debug_set_param(F.TM_DBG_UPLOAD_STREAMS, eng_str)
    keep, desc = [], []
    # round 6: in a quarter of the cases the page-locked pictures of a side lie back to back in ONE pool, slot after slot (a ring, a surface pool):
    # the engine then sends neighbours up as one DMA (merge limit drawn too: off, pairs, everything)
    pooled = rng.random() < 0.25
    pools = None
    if pooled:
        eng.debug_set_param(F.TM_DBG_UPLOAD_MERGE, int(rng.choice([0, 1 << 20, 14 << 20, 1 << 30])))
        bps_ = 1 if bits == 8 else 2
        nb = (w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)) * bps_
        pools = [torch.empty(nb * B + 64, dtype=torch.uint8).pin_memory() for _ in range(2)]
    for slot in range(B):
        planes = tm.synth.yuv420_pair(w, h, int(rng.integers(0, 500)), bits if bits != 16 else 10)
        for side in range(2):
            Y, Cb, Cr = planes[side]
            if bits == 16:  # full 16-bit content: the 10-bit pattern scaled up
                Y, Cb, Cr = (Y.astype(np.uint32) * 64 + 17).astype(np.uint16), (Cb.astype(np.uint32) * 64 + 5).astype(np.uint16), (Cr.astype(np.uint32) * 64 + 9).astype(np.uint16)
            pb = 8 if bits == 8 else bits
            sb, pit, ch = tm.synth.pack_biplanar((Y, Cb, Cr), w, h, pb)
            mkb = tm.HwFrame.nv12 if bits == 8 else tm.HwFrame.p016
            base.set_frame(slot, side, mkb(torch.from_numpy(sb).cuda(), pit, ch))
            mem = str(rng.choice(["device", "pinned", "host"]))
            how = str(rng.choice(["surface", "i420", "i420_tight"] + (["p10", "p10_tight"] if bits == 10 else [])))
            if pooled and rng.random() < 0.85:  # (now and then a frame of another form in between: it breaks the run, nothing else)
                mem, how = "pinned", "i420_tight"
            if how in ("p10", "p10_tight"):  # round 6: the same 10-bit planes packed three samples to a word, rows padded or the whole picture tight
                cw, chh = (w + 1) // 2, (h + 1) // 2
                wy, wc = tm.synth.p10_row_words(w), tm.synth.p10_row_words(cw)
                if how == "p10_tight":
                    flat = np.concatenate([tm.synth.p10_pack_plane(p).ravel() for p in (Y, Cb, Cr)]).view(np.int32)
                    buf = place(flat, mem); keep.append(buf)
                    cut = (lambda a, r, c: a.reshape(r, c)) if mem == "host" else (lambda a, r, c: a.view(r, c))
                    y, u, v = cut(buf[: h * wy], h, wy), cut(buf[h * wy: h * wy + chh * wc], chh, wc), cut(buf[h * wy + chh * wc:], chh, wc)
                else:
                    pad_c = 2 * int(rng.integers(0, 20))
                    pk = lambda p, pad: (lambda t: t[:, : t.shape[1] - pad] if pad else t)(place(tm.synth.p10_pack_plane(p, tm.synth.p10_row_words(p.shape[1]) + pad).view(np.int32), mem))
                    y, u, v = pk(Y, 2 * int(rng.integers(0, 20))), pk(Cb, pad_c), pk(Cr, pad_c)
                fr = tm.HwFrame.i420p10(y, u, v)
            elif how == "surface":
                bps = 1 if bits == 8 else 2
                pitch = (max(w, 2 * ((w + 1) // 2)) + int(rng.integers(0, 150))) * bps
                coded = h + int(rng.integers(0, 40))
                s2, p2, c2 = tm.synth.pack_biplanar((Y, Cb, Cr), w, h, pb, pitch=pitch, coded_height=coded)
                fr = mkb(place(s2, mem), p2, c2)
            else:
                dt = np.uint8 if bits == 8 else np.uint16
                if how == "i420_tight":  # one allocation: Y rows, then Cb rows, then Cr rows, no padding
                    flat = np.concatenate([Y.astype(dt).ravel(), Cb.astype(dt).ravel(), Cr.astype(dt).ravel()])
                    if pooled and mem == "pinned":
                        raw = pools[side][slot * nb:(slot + 1) * nb]
                        raw.copy_(torch.from_numpy(flat.view(np.uint8)))
                        buf = raw if bits == 8 else raw.view(torch.int16)
                    else:
                        buf = place(flat, mem)
                    cw, chh = (w + 1) // 2, (h + 1) // 2
                    if mem == "host":
                        y, u, v = buf[: w * h].reshape(h, w), buf[w * h: w * h + cw * chh].reshape(chh, cw), buf[w * h + cw * chh:].reshape(chh, cw)
                    else:
                        y, u, v = buf[: w * h].view(h, w), buf[w * h: w * h + cw * chh].view(chh, cw), buf[w * h + cw * chh:].view(chh, cw)
                    keep.append(buf)
                else:
                    def padded(p, pad):
                        q = np.zeros((p.shape[0], p.shape[1] + pad), dt)
                        q[:, : p.shape[1]] = p
                        t = place(q, mem)
                        return t[:, : p.shape[1]]
                    pad_c = int(rng.integers(0, 40))  # (Cb and Cr share one pitch: tm_engine_set_frame_i420 takes pitch_uv once)
                    y, u, v = padded(Y, int(rng.integers(0, 40))), padded(Cb, pad_c), padded(Cr, pad_c)
                fr = tm.HwFrame.i420(y, u, v, bits=bits)
            keep.append(fr)
            desc.append((slot, side, how, mem))
            try:
                eng.set_frame(slot, side, fr)
            except Exception as ex:
                print(f"EXCEPTION case {case}: {w}x{h} {bits}-bit {how} {mem}: {ex}", flush=True)
                raise
            if rng.random() < 0.3:
                tok = eng.upload_fence()
                if rng.random() < 0.5:
                    eng.upload_done(tok, block=True)
    torch.cuda.synchronize()
    base.compute_async(B); base.sync()
    for rep in range(2):
        eng.compute_async(B); eng.sync()
        for i in range(B):
            if not (np.array_equal(eng.raw_sums(i), base.raw_sums(i)) and eng.sse(i) == base.sse(i)):
                bad += 1
                print(f"MISMATCH case {case}: {w}x{h} {bits}-bit batch {B} slot {i} rep {rep}: {[d for d in desc if d[0] == i]} linear={eng_lin} streams={eng_str}", flush=True)
    eng.close(); base.close()
print(f"surface sweep: {cases} random cases (every hand-over form, memory kind, upload mode), mismatches {bad}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
