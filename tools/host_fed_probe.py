#!/usr/bin/env python3
"""GPU box: the host-fed leg of bench.py alone (page-locked host -> H2D every step, two engines ping-pong), per workload and batch.
usage: host_fed_probe.py [workload:batch,...]   (tuning values: tm_engine_debug_set_param on the engines run_host_fed creates)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
cases = sys.argv[1] if len(sys.argv) > 1 else "1080p_nv12:32,4k_p016:8"
sys.argv = sys.argv[:1]
args = bench.parse_args()
ctx = bench.Ctx(args)
ctx.tm.init_hip(0)
for case in cases.split(","):
    wl, B = case.split(":")
    r = bench.run_host_fed(ctx, args, wl, int(B), 40, 2)
    print(json.dumps({"case": case, "pairs_per_s": round(r["value"], 1), "ms_per_step": round(r["ms_per_step"], 3), "h2d_GBs": round(r["h2d_GBs_per_gpu"], 1)}), flush=True)
