#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: mean counter value per kernel launch.
Counter unit is KiB (rocprofv3 derived metric); gfx950 corrections are applied by the reader
(MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reports wide coalesced streaming reads by 2x)."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16  # the profile is stamped with the kernel sources it was taken with (bench.py refuses another version's)


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return acc


fd, wd, wl = sys.argv[1], sys.argv[2], sys.argv[3]
fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
out = {"workload": wl, "unit": "KiB per launch (raw counter)", "csrc_sha16": csrc_sha16(), "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    f = fe.get(k, []); w = wr.get(k, [])
    # skip the first (warm-up) launch of each kernel
    f = f[1:] or f; w = w[1:] or w
    out["kernels"][k] = {"launches": max(len(f), len(w)), "FETCH_SIZE": sum(f) / len(f) if f else None,
                         "WRITE_SIZE": sum(w) / len(w) if w else None}
print(json.dumps(out, indent=1))
