#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: the timeline of the last complete steps of bench.py -- per kernel launch its queue, start and
end relative to the step's first kernel -- i.e. which kernels ran at the same time.  usage: trace_timeline.py KERNEL_TRACE.csv [steps]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tmk::" in r["Kernel_Name"] and ", 1>" not in r["Kernel_Name"]]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_ingest_rows" in r["Kernel_Name"] or "k_ingest_wave" in r["Kernel_Name"]]
all_steps = list(zip(starts, starts[1:] + [len(rows)]))
mid = len(all_steps) // 2  # the middle of the run: the timed steps (the last ones are the alone-mode leg and the compare leg)
for s, e in all_steps[mid:mid + steps] + all_steps[-2:-1]:
    t0 = int(rows[s]["Start_Timestamp"])
    print(f"step of {e - s} launches, {(max(int(r['End_Timestamp']) for r in rows[s:e]) - t0) / 1e6:.3f} ms from the first start to the last end")
    for r in rows[s:e]:
        name = r["Kernel_Name"].replace("void ", "").replace("tmk::", "").split("(")[0]
        print(f"  queue {r['Queue_Id']:>2}  {(int(r['Start_Timestamp']) - t0) / 1e6:7.3f} .. {(int(r['End_Timestamp']) - t0) / 1e6:7.3f} ms  {name[:70]}")
