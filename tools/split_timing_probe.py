#!/usr/bin/env python3
"""HISTORICAL (round 4): needs the -DTM_SPLIT_TIMING lab build (`make -C turbo-metrics_amd/csrc exp`) of commit 3773fd2; the product sources
carry no lab switches since round 5.  Its output is profiles/r04r_ / r04s_split_timing.log.

GPU box, lab build (make -C turbo-metrics_amd/csrc exp; TM_HIP_LIB=build_exp/libturbometrics_hip.so): which wave of the multi-wave row
pass does a phase wait for?  Per wave of the first row block (scale 0, channel Y, rows 0-63) of a one-pair launch: shader cycles of WORK
between barriers, per phase and per step, against the kernel's total.  usage: TM_HIP_LIB=... split_timing_probe.py [pairs]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm
assert "build_exp" in tm.ffi.LIB_PATH, "point TM_HIP_LIB at the lab build"
L = tm.ffi.lib()
tm.init_hip(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, h = 1920, 1080
eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
(rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, 1)
rt, dt = torch.from_numpy(rs).cuda(), torch.from_numpy(ds).cuda()
for slot in range(B):
    eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
for _ in range(200):
    eng.compute_async(B); eng.sync()
out = (C.c_ulonglong * 32)()
L.tm_debug_read_split_timing.argtypes = [C.POINTER(C.c_ulonglong)]
assert L.tm_debug_read_split_timing(out) == 0
names = ["sigma11", "sigma22", "sigma12", "mu1", "mu2", "ref/dis fetch", "ssim consumer", "edge consumer"]
steps = w + 4
print(f"{B} pair(s) per launch, first row block of scale 0 / Y: {steps} steps")
for wv in range(8):
    work, n, total = out[4 * wv], out[4 * wv + 1], out[4 * wv + 2]
    print(f"  wave {wv} {names[wv]:14s}: work {work:8d} cycles = {work / max(n, 1):7.0f} per phase = {work / steps:6.1f} per step; {n} phases; alive {total} cycles ({100.0 * work / max(total, 1):.0f} % working)")
eng.close()
