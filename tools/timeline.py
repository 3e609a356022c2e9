#!/usr/bin/env python3
"""Print a kernel timeline (start/end in µs relative to a reference kernel) from a rocprofv3 --kernel-trace CSV.
usage: timeline.py <kernel_trace.csv> [first_row] [n_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
def short(name):
    for k in ("k_step_begin", "k_step_filter", "k_step_resolve", "k_step_regression", "k_step_finish", "k_posterior_eigen_rr", "k_posterior_eigen"):
        if k in name: return k
    name = name.replace("icp::(anonymous namespace)::", "").replace("icp::tri::", "tri::").replace("icp::", "").replace("void ", "")
    return name.split("(")[0][:40]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %7.1f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
