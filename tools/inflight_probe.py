#!/usr/bin/env python3
"""GPU box: small launches with several of them in flight -- N engines (own arenas, own streams, one shared side stream) take turns,
an engine is only waited for right before its next launch.  One pair per launch leaves most of the chip idle (111 row blocks on 256
CUs): how far does the rate go with 1 / 2 / 3 / 4 / 6 launches in flight?  Scores identical on every engine.
usage: inflight_probe.py [workload] [pairs per launch, comma separated] [engines, comma separated]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "1080p_nv12"
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
depths = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3,4,6").split(",")]
sys.argv = sys.argv[:1]
args = bench.parse_args()
ctx = bench.Ctx(args)
tm, torch = ctx.tm, ctx.torch
w, h, kind, _, _ = bench.WORKLOADS[wl]
tm.set_placement_candidates(1)
for B in batches:
    engs = [tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B) for _ in range(max(depths))]
    for e in engs:
        ctx.fill_slots(e, wl, min(32, B), 0, B)
    torch.cuda.synchronize()
    row = {"workload": wl, "pairs_per_launch": B}
    for n in depths:
        def run(k):
            busy = [False] * n
            for i in range(k):
                e = engs[i % n]
                if busy[i % n]:
                    e.sync()
                e.compute_async(B); busy[i % n] = True
            for i in range(n):
                if busy[i]:
                    engs[i].sync()
        run(max(20, 200 // B))
        k = max(60, int(0.5 / (0.00033 + 0.00007 * B) / 1))
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(k); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row[f"in_flight_{n}"] = round(B * k / dt)
        s0 = [s.ssimulacra2 for s in engs[0].scores_batch(B)]
        for e in engs[1:n]:
            assert s0 == [s.ssimulacra2 for s in e.scores_batch(B)]
    print(json.dumps(row), flush=True)
    for e in engs:
        e.close()
