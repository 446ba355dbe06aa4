#!/usr/bin/env python3
"""GPU box: compute_one_deferred / collect at 2 ... 8 pairs in flight (set_deferred_depth), frames in HBM, one fresh process per depth
(the runtime hands its hardware queues out by the order in which streams were created: a process that has created and freed other
engines before measures another mapping).  usage: deferred_depth_probe.py [depths, comma separated] [WxH]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from tm_pkg import tm
    depth, w, h = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    tm.init_hip(0)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    frames = []
    for n in range(4):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
        frames.append((tm.HwFrame.nv12(torch.from_numpy(rs).cuda(), rp, rch), tm.HwFrame.nv12(torch.from_numpy(ds).cuda(), dp, dch)))
    torch.cuda.synchronize()
    want = [eng.compute_one(*f).ssimulacra2 for f in frames]
    if depth == 1:
        def run(k):
            return [eng.compute_one(*frames[i % 4]).ssimulacra2 for i in range(k)]
    else:
        eng.set_deferred_depth(depth, create_now=True)

        def run(k):
            got, tickets = [], []
            for i in range(k):
                tickets.append(eng.compute_one_deferred(*frames[i % 4]))
                if len(tickets) >= depth:
                    got.append(eng.collect(tickets.pop(0)).ssimulacra2)
            return got + [eng.collect(t).ssimulacra2 for t in tickets]
    run(200)
    t0 = time.perf_counter(); got = run(3000); dt = time.perf_counter() - t0
    assert got[:4] == want
    print(json.dumps({"depth": depth, "size": f"{w}x{h}", "pairs_per_s": round(3000 / dt)}), flush=True)
    sys.exit(0)
depths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,6,8").split(",")]
w, h = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
for d in depths:
    r = subprocess.run([sys.executable, __file__, "--one", str(d), str(w), str(h)], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-400:], flush=True)
