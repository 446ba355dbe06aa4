import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tm_pkg import tm
tm.init_hip(0)
for case in sys.argv[1].split(","):
    size,B=case.split(":"); w,h=(int(v) for v in size.split("x")); B=int(B); p016=w>3000
    pairs=[]
    for n in range(min(B,4)):
        (rs,rp,rch),(ds,dp,dch)=(tm.synth.p016_pair if p016 else tm.synth.nv12_pair)(w,h,n)
        pairs.append(((torch.from_numpy(rs).cuda(),rp,rch),(torch.from_numpy(ds).cuda(),dp,dch)))
    eng=tm.TurboMetrics(w,h,tm.Metrics(ssimulacra2=True),batch=B)
    mk=tm.HwFrame.p016 if p016 else tm.HwFrame.nv12
    for s in range(B):
        (rt,rp,rch),(dt,dp,dch)=pairs[s%len(pairs)]; eng.set_pair(s,mk(rt,rp,rch),mk(dt,dp,dch))
    row={"case":case}
    for name,var in (("two_pass",tm.ffi.TM_VARIANT_TWO_PASS_EDGE),("default",tm.ffi.TM_VARIANT_DEFAULT),("fused",tm.ffi.TM_VARIANT_FUSED_EDGE)):
        eng.set_variant(var)
        for _ in range(10): eng.compute_async(); eng.sync()
        t0=time.perf_counter()
        for _ in range(100): eng.compute_async(); eng.sync()
        row[name]=round((time.perf_counter()-t0)/100*1e3,4)
    row["default_is_fused"]=eng.uses_fused_edge() if False else None
    print(json.dumps(row),flush=True)
    eng.close()
