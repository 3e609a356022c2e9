#!/usr/bin/env python3
"""a window of a rocprofv3 kernel trace as a timeline: queue, start, end (µs from the window's first kernel), name.
usage: r6_trace_window.py <kernel_trace.csv> <first kernel index as a fraction of the trace> <count>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
a = int(float(sys.argv[2]) * len(rows)); n = int(sys.argv[3])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:a + n]:
    m = re.search(r"k_\w+", r["Kernel_Name"])
    print("q%-2s %8.1f %8.1f  %s" % (r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, m.group(0) if m else r["Kernel_Name"][:28]))
