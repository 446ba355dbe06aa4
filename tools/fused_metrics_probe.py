import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tm_pkg import tm
tm.init_hip(0)
for (w,h,B,p016) in ((1920,1080,64,False),(3840,2160,24,True),(1920,1080,16,False)):
    pairs=[]
    for n in range(4):
        (rs,rp,rch),(ds,dp,dch) = (tm.synth.p016_pair if p016 else tm.synth.nv12_pair)(w,h,n)
        pairs.append(((torch.from_numpy(rs).cuda(),rp,rch),(torch.from_numpy(ds).cuda(),dp,dch)))
    eng = tm.TurboMetrics(w,h,tm.Metrics(ssimulacra2=True,psnr=True,msssim=True),batch=B)
    mk = tm.HwFrame.p016 if p016 else tm.HwFrame.nv12
    for s in range(B):
        (rt,rp,rch),(dt,dp,dch)=pairs[s%4]; eng.set_pair(s, mk(rt,rp,rch), mk(dt,dp,dch))
    for _ in range(5): eng.compute_async(); eng.sync()
    t0=time.perf_counter()
    for _ in range(30): eng.compute_async(); eng.sync()
    dt=(time.perf_counter()-t0)/30*1e3
    sc=[(eng.scores(i).ssimulacra2, eng.scores(i).msssim, eng.scores(i).psnr) for i in range(min(B,4))]
    print(json.dumps({"case": f"{w}x{h}:{B}", "ms": round(dt,3), "scores": sc[:2]}), flush=True)
    eng.close()
