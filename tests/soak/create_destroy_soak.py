#!/usr/bin/env python3
"""GPU box: engines created, run and destroyed in one process, device memory read through the C ABI (tm_device_mem_info =
hipMemGetInfo) -- no torch in the process, so no caching allocator between the measurement and the driver (round 3's soak compared
torch.cuda.mem_get_info before and after while torch still held the test's device tensors in its cache: its "leak MiB 412.0").
200 cycles over several sizes, metric masks and batch sizes, every cycle with the fused kernel's side stream and hand-off buffers,
every fourth with a second engine alive at the same time; must end within 16 MiB of where it stood after the first cycle (the first
engine leaves the runtime's own pools and code objects behind).  usage: create_destroy_soak.py [cycles]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
from tm_pkg import tm
assert "torch" not in sys.modules or os.environ.get("TM_SOAK_ALLOW_TORCH"), "this probe must run without torch in the process"
L = tm.ffi.lib()
tm.init_hip(0)
tm.set_placement_candidates(1)


def free_mib():
    f, t = C.c_size_t(), C.c_size_t()
    assert L.tm_device_mem_info(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20


cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = [(1920, 1080, 8), (640, 360, 16), (1280, 720, 6), (3840, 2160, 3), (333, 203, 2)]
frames = {}
base = None
start = free_mib()
for i in range(cycles):
    w, h, B = cases[i % len(cases)]
    m = tm.Metrics(ssimulacra2=True, psnr=(i % 2 == 0), msssim=(i % 3 == 0 and min(w, h) >= 176))
    if (w, h) not in frames:
        frames[(w, h)] = tm.synth.nv12_pair(w, h, 1)
    (rs, rp, rch), (ds, dp, dch) = frames[(w, h)]
    engs = [tm.TurboMetrics(w, h, m, batch=B) for _ in range(2 if i % 4 == 3 else 1)]
    for eng in engs:
        eng.set_variant(tm.ffi.TM_VARIANT_FUSED_EDGE)  # side stream, hand-off buffers, status words: every cycle
        for slot in range(B):
            eng.set_pair(slot, tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
        eng.compute_async(B); eng.sync()
        s = eng.scores(B - 1).ssimulacra2
        if i % 7 == 0:
            eng.set_full_sums(True); eng.compute_async(B); eng.sync(); eng.set_full_sums(False)
    for eng in engs:
        eng.close()
    now = free_mib()
    if base is None:
        base = now
    if i % 20 == 0 or i == cycles - 1:
        print(f"cycle {i}: {w}x{h} x{B} score {s:.6f}, free {now:.0f} MiB ({base - now:+.1f} MiB against the end of cycle 0)", flush=True)
end = free_mib()
print(f"before the first engine {start:.0f} MiB free, after cycle 0 {base:.0f}, after cycle {cycles - 1} {end:.0f}: leak {base - end:.1f} MiB over {cycles - 1} cycles")
assert abs(base - end) <= 16.0, "device memory did not come back"
print("ok")
