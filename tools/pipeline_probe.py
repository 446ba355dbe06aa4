#!/usr/bin/env python3
"""GPU box: two batches in flight.  Engines A and B (own arenas, own streams) take turns: while A's blur region runs,
B's ingest stage is already on the device.  `serial` = one engine, compute_async + sync per step; `two_engines` = both engines kept
busy (sync of engine X only right before its next submit).  Same frames, scores must be identical.
usage: pipeline_probe.py [workload] [batch per engine] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "1080p_nv12"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
mets = sys.argv[4] if len(sys.argv) > 4 else "ssimulacra2"
sys.argv = sys.argv[:1]
args = bench.parse_args()
ctx = bench.Ctx(args)
tm, torch = ctx.tm, ctx.torch
w, h, kind, _, _ = bench.WORKLOADS[wl]
distinct = min(32 if w * h <= 1920 * 1080 else 2, B)
M = tm.Metrics(ssimulacra2="ssimulacra2" in mets, psnr="psnr" in mets, msssim="msssim" in mets)
engs = [tm.TurboMetrics(w, h, M, batch=B) for _ in range(2)]
for e in engs:
    ctx.fill_slots(e, wl, distinct, 0, B)
torch.cuda.synchronize()


def serial(e, k):
    for _ in range(k):
        e.compute_async(B); e.sync()


def two(k):
    busy = [False, False]
    for i in range(k):
        e = engs[i & 1]
        if busy[i & 1]:
            e.sync()
        e.compute_async(B)
        busy[i & 1] = True
    for i, e in enumerate(engs):
        if busy[i]:
            e.sync()


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    serial(engs[0], 1)
out = {"workload": wl, "batch_per_engine": B, "steps": steps, "metrics": mets}
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); serial(engs[0], steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.setdefault("serial_pairs_per_s", []).append(round(B * steps / dt, 1))
    s0 = [s.ssimulacra2 for s in engs[0].scores_batch(B)]
    torch.cuda.synchronize(); t0 = time.perf_counter(); two(steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.setdefault("two_engines_pairs_per_s", []).append(round(B * steps / dt, 1))
    assert s0 == [s.ssimulacra2 for s in engs[1].scores_batch(B)] == [s.ssimulacra2 for s in engs[0].scores_batch(B)]
# shaped: A's ingest beside B's row pass and the other way round (tm_engine_debug_chain)
engs[0].debug_chain(engs[1]); engs[1].debug_chain(engs[0])
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); two(steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.setdefault("two_engines_chained_pairs_per_s", []).append(round(B * steps / dt, 1))
    assert s0 == [s.ssimulacra2 for s in engs[1].scores_batch(B)] == [s.ssimulacra2 for s in engs[0].scores_batch(B)]
engs[0].debug_chain(None); engs[1].debug_chain(None)
print(json.dumps(out), flush=True)
for e in engs:
    e.close()
