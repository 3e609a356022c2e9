#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats summary (…kernel_stats.csv) -> the markdown table kept under profiles/.
usage: stats_md.py <kernel_stats.csv> "<command line>" > profiles/<name>.md"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats summary\n")
print("command: `%s`\n" % sys.argv[2])
print("| kernel | calls | avg us | min us | max us | total ms | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"])
    print("| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.1f |" % (r["Name"][:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                            float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
print("\ntotal kernel time: %.2f ms" % (tot / 1e6))
