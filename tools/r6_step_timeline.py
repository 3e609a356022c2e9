#!/usr/bin/env python3
"""one step of the wide on-device loop as a timeline (rocprofv3 kernel trace): every kernel between two k_mhw_front launches of the
middle of the run — queue, start and end relative to the step's first kernel (µs), name.  usage: r6_step_timeline.py <kernel_trace.csv> [which]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fr = [i for i, r in enumerate(rows) if "k_mhw_front" in r["Kernel_Name"]]
w = int(sys.argv[2]) if len(sys.argv) > 2 else len(fr) // 2
a, b = fr[w], fr[w + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    m = re.search(r"k_\w+", r["Kernel_Name"])
    print("q%-2s %8.1f %8.1f  %-28s grid %s" % (r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                          m.group(0) if m else r["Kernel_Name"][:28], r.get("Grid_Size", "?")))
