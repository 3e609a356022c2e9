#!/usr/bin/env python3
"""Merged timeline of kernels and HIP API calls (rocprofv3 --kernel-trace --hip-runtime-trace CSVs), µs relative to a reference kernel.
usage: timeline2.py <kernel_trace.csv> <hip_api_trace.csv> [first_kernel_row] [span_us]"""
import csv, sys
k = list(csv.DictReader(open(sys.argv[1])))
a = list(csv.DictReader(open(sys.argv[2])))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
span = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 3e6
t0 = int(k[first]["Start_Timestamp"])
def short(name):
    name = name.replace("icp::(anonymous namespace)::", "").replace("icp::tri::", "tri::").replace("icp::", "").replace("void ", "")
    return name.split("(")[0][:44]
ev = []
for r in k:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if 0 <= s <= span: ev.append((s, e, "  GPU q%-2s %s" % (r.get("Queue_Id", "?"), short(r["Kernel_Name"]))))
for r in a:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if 0 <= s <= span: ev.append((s, e, "host t%s %s" % (r.get("Thread_Id", "?")[-3:], r["Function"])))
ev.sort()
for s, e, w in ev:
    print("%9.1f %9.1f %7.1f  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, w))
