#!/usr/bin/env python3
"""GPU box: twelve engines of varying metric masks and batch sizes created, run and destroyed in one process -- device memory must
come back every time (the first engine's ~150 MiB of runtime pools stay)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from tm_pkg import tm
tm.init_hip(0)
free0 = torch.cuda.mem_get_info()[0]
for i in range(12):
    m = tm.Metrics(ssimulacra2=True, psnr=(i % 2 == 0), msssim=(i % 3 == 0))
    eng = tm.TurboMetrics(1920, 1080, m, batch=64 if i % 4 else 16)
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(1920, 1080, i)
    eng.set_pair(0, tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
    eng.compute_async(1); eng.sync()
    s = eng.scores(0).ssimulacra2
    eng.close()
    print(i, round(s, 6), "free GiB", round(torch.cuda.mem_get_info()[0] / 2**30, 2), flush=True)
free1 = torch.cuda.mem_get_info()[0]
print("leak MiB", (free0 - free1) / 2**20)
