#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: the C ABI under random (mostly invalid) arguments, through ctypes -- slots and sides out of range, null and misaligned
pointers, pitches below a row / above 2^24 / overflowing 4 GB, bit depths out of range, reads before any compute, launches of unset
slots, tokens never issued, engines of impossible sizes.  Every call must RETURN (a TM_* code), none may crash the process, and
afterwards the engine must still produce the right scores.  usage: abi_fuzz_soak.py [calls]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tm_pkg import tm
F = tm.ffi
L = F.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tm.init_hip(0); tm.set_placement_candidates(1)
rng = np.random.default_rng(99)
w, h, B = 200, 120, 3
eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True, ssim=True), batch=B)
hnd = eng._h
(rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, 3)
good = (tm.HwFrame.nv12(torch.from_numpy(rs).cuda(), rp, rch), tm.HwFrame.nv12(torch.from_numpy(ds).cuda(), dp, dch))
want = eng.compute_one(*good)
dev = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
host = np.zeros(1 << 20, np.uint8)
codes = {}
VALID = set(range(6))


def ptr():
    k = int(rng.integers(0, 6))
    if k == 0: return None
    if k == 1: return C.c_void_p(dev.data_ptr() + int(rng.integers(0, 64)))
    if k == 2: return C.c_void_p(host.ctypes.data + int(rng.integers(0, 64)))
    if k == 3: return C.c_void_p(1)  # (only ever passed with arguments that are refused before it is read, or as TM_MEM_DEVICE and never launched)
    return C.c_void_p(dev.data_ptr())


def pitch():
    return int(rng.choice([0, 1, w - 1, w, w * 2, 4096, (1 << 24) - 1, 1 << 24, 1 << 31, (1 << 32) + 5, (1 << 40)]))


def note(name, rc):
    assert rc in VALID or rc < 0, (name, rc)
    codes[(name, rc)] = codes.get((name, rc), 0) + 1


t0 = time.time()
for i in range(N):
    k = int(rng.integers(0, 12))
    slot, side = int(rng.choice([0, 1, 2, 3, 7, 2 ** 31 - 1, 2 ** 32 - 1])), int(rng.choice([0, 1, 2, -1, 99]))
    mem = int(rng.choice([0, 1, 2, 3, -1]))
    matrix, transfer, full = int(rng.choice([0, 1, 2, 3, -1])), int(rng.choice([0, 0, 1])), int(rng.choice([0, 0, 1]))
    p0, p1, p2 = ptr(), ptr(), ptr()
    if p0 is not None and p0.value == 1 and mem != 1: p0 = None  # a wild pointer is never handed over as host memory (it would be read)
    if p1 is not None and p1.value == 1 and mem != 1: p1 = None
    if p2 is not None and p2.value == 1 and mem != 1: p2 = None
    pt = pitch()
    if mem in (0, 2) and pt * (h + h // 2 + 1) > (1 << 19):  # host copies read pitch * rows bytes of OUR buffer: keep them inside it
        pt = w
    if k == 0: note("set_frame_nv12", L.tm_engine_set_frame_nv12(hnd, slot, side, p0, p1, pt, matrix, transfer, full, mem))
    elif k == 1: note("set_frame_p016", L.tm_engine_set_frame_p016(hnd, slot, side, p0, p1, pt, matrix, transfer, full, mem))
    elif k == 2: note("set_surface_nv12", L.tm_engine_set_surface_nv12(hnd, slot, side, p0, pt, int(rng.choice([0, h - 1, h, h + 8, 2 ** 31])) if mem == 1 else h, matrix, transfer, full, mem))
    elif k == 3 and rng.random() < 0.4:  # round 6: the packed 10-bit kind (pointers and pitches must be multiples of 8, a row is whole 512-byte blocks)
        note("set_frame_i420p10", L.tm_engine_set_frame_i420p10(hnd, slot, side, p0, p1, p2, pitch() if mem == 1 else int(rng.choice([512, 512, 520, 256, 1024])),
                                                                pitch() if mem == 1 else int(rng.choice([512, 512, 516, 0])), matrix, transfer, full, mem))
    elif k == 3: note("set_frame_i420", L.tm_engine_set_frame_i420(hnd, slot, side, p0, p1, p2, pt, pitch() if mem == 1 else w, int(rng.choice([0, 7, 8, 10, 16, 17, -3])), matrix, transfer, full, mem))
    elif k == 4: note("set_frame_rgb8", L.tm_engine_set_frame_rgb8(hnd, slot, side, p0, pt if mem == 1 else w * 3, mem))
    elif k == 5: note("set_frame_rgbf32", L.tm_engine_set_frame_rgbf32(hnd, slot, side, p0, pt if mem == 1 else w * 12, mem))
    elif k == 6:
        s = F.FrameScoresC()
        note("get_scores", L.tm_engine_get_scores(hnd, slot, C.byref(s)))
    elif k == 7:
        out = (C.c_double * 108)()
        note("get_raw_sums", L.tm_engine_get_raw_sums(hnd, slot, out))
    elif k == 8: note("upload_done", L.tm_engine_upload_done(hnd, int(rng.choice([0, 1, 10 ** 6, 2 ** 63])), 0))
    elif k == 9: note("compute_async_bad_n", L.tm_engine_compute_async(hnd, int(rng.choice([0, B + 1, 2 ** 31]))))
    elif k == 10:
        e2 = C.c_void_p()
        rc = L.tm_engine_create(C.byref(e2), int(rng.choice([0, 1, 5, 16385, 2 ** 31])), int(rng.choice([0, 1, 5, 16385])), int(rng.choice([0, 1, 8, 15, 16, 255])), int(rng.choice([0, 1, 2])))
        note("create", rc)
        if rc == 0: L.tm_engine_destroy(e2)
    else:
        note("set_variant", L.tm_engine_set_variant(hnd, int(rng.choice([0, 1, 2, 3, 0x100, 0x5000, 0x400, -1, 2 ** 20]))))
        L.tm_engine_set_variant(hnd, 0)
    if i % 500 == 499:  # the engine still works: the good pair again, bit for bit
        got = eng.compute_one(*good)
        assert got == want, (i, got, want)
print(f"abi fuzz: {N} random calls in {time.time() - t0:.0f} s, no crash, scores unchanged at every check; return codes seen:")
for (name, rc), n in sorted(codes.items()):
    print(f"   {name:22s} rc {rc:3d}: {n}")
