#!/usr/bin/env python3
"""GPU box: launches of 1 .. 16 pairs of 1080p (the reference's compute_one is ONE pair per call) : the default choice of kernels
against the eight-wave row pass forced / forbidden, the EDGE jobs in the two passes, hipGraph replay.
Per batch and configuration: wall ms per step (compute_async + sync), stage ms, pairs/s; scores must not change.
usage: small_launch_probe.py [batches] [WxH]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tm_pkg import tm
F = tm.ffi
w, h = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
tm.init_hip(0)
tm.set_placement_candidates(1)
gen = tm.synth.nv12_pair
pairs = []
for n in range(4):
    (rs, rp, rch), (ds, dp, dch) = gen(w, h, n)
    pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
BIG = 1 << 40
SEL = os.environ.get("PROBE_CONFIGS")  # comma list of config names (default: all)
CONFIGS = [  # name, {param: value}, graph
    ("default", {}, False),
    ("solo_col", {F.TM_DBG_SOLO_COL_BELOW: BIG}, False),
    ("solo_col+graph", {F.TM_DBG_SOLO_COL_BELOW: BIG}, True),
    ("split_forced", {"variant": F.TM_VARIANT_SPLIT_ROWS}, False),
    ("whole_rows", {"variant": F.TM_VARIANT_WHOLE_ROWS}, False),
    ("two_pass_edge", {"variant": F.TM_VARIANT_TWO_PASS_EDGE}, False),
    ("fused_edge", {"variant": F.TM_VARIANT_FUSED_EDGE}, False),
    ("graph", {}, True),
]
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,6,8,12,16").split(",")]:
    sc = None
    for name, params, graph in CONFIGS:
        if SEL and name not in SEL.split(","):
            continue
        eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
        for slot in range(B):
            (rt, rp, rch), (dt, dp, dch) = pairs[slot % 4]
            eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
        for k, v in params.items():
            if k == "variant":
                eng.set_variant(v)
            else:
                eng.debug_set_param(k, v)
        eng.set_graph(graph)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.08:
            eng.compute_async(); eng.sync()
        k = 300 if B <= 4 else 150
        t0 = time.perf_counter()
        for _ in range(k):
            eng.compute_async(); eng.sync()
        wall = (time.perf_counter() - t0) / k * 1e3
        stages = None
        if not graph:
            eng.set_profiling(True); eng.stage_ms(reset=True)
            for _ in range(60):
                eng.compute_async(); eng.sync()
            ms, n = eng.stage_ms(reset=True)
            stages = [round(m / n, 4) for m in ms]
            eng.set_profiling(False)
        got = [eng.scores(i).ssimulacra2 for i in range(B)]
        assert sc is None or sc == got, (name, B)
        sc = got
        print(json.dumps({"batch": B, "config": name, "wall_ms": round(wall, 4), "pairs_per_s": round(B / wall * 1e3), "fused_edge": eng.uses_fused_edge(B),
                          "stage_ms[ingest,col,row,finish,edge]": stages}), flush=True)
        eng.close()
