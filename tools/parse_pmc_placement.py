#!/usr/bin/env python3
"""rocprofv3 --pmc output of tools/placement_debug.py: duration and counter values of the placement search's dispatches, per
candidate and per pass (the last 3 x N dispatches of k_blur_v_jobs<32,16,1> / k_blur_h_jobs_x<...,1> are the three timing rounds
over the N candidates, in order)."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for tag, name in (("column pass", "k_blur_v_jobs<32, 16, 1>"), ("row pass", "k_blur_h_jobs_x<16, 8, 32, 16, 1>")):
    rows, dur = defaultdict(dict), {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                i = int(r["Dispatch_Id"])
                rows[i][r["Counter_Name"]] = float(r["Counter_Value"])
                dur[i] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ids = sorted(rows)[-3 * N:]
    if len(ids) < 3 * N: continue
    print(tag)
    print("  %-40s" % "duration ms (fastest of 3)", " ".join("c%d:%.3f" % (c, min(dur[ids[rnd * N + c]] for rnd in range(3))) for c in range(N)))
    for n in sorted({k for i in ids for k in rows[i]}):
        print("  %-40s" % n, " ".join("c%d:%.4g" % (c, sum(rows[ids[rnd * N + c]].get(n, float("nan")) for rnd in range(3)) / 3) for c in range(N)))
