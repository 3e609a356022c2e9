#!/usr/bin/env python3
"""Per-kernel duration statistics (min / median / mean, µs) of a rocprofv3 --kernel-trace CSV."""
import csv, sys, statistics, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k in ("k_step_begin_batch", "k_step_begin", "k_step_filter", "k_step_resolve", "k_step_regression", "k_step_finish", "k_posterior_eigen_rr"):
        if k in n:
            n = k
            break
    else:
        n = n.split("(")[0][-40:]
    acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-28s n=%5d  min %7.2f  p10 %7.2f  med %7.2f  mean %7.2f  total %9.1f" % (n, len(v), v[0], v[len(v) // 10], statistics.median(v), sum(v) / len(v), sum(v)))
