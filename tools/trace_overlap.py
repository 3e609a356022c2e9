#!/usr/bin/env python3
"""overlap of the hardware queues in a rocprofv3 kernel trace: per queue the busy time, the union over queues, the wall span
usage: trace_overlap.py <kernel_trace.csv> [skip_fraction]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
per = collections.defaultdict(float); names = collections.defaultdict(lambda: collections.defaultdict(float))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    per[q] += e - s
    km = re.search(r"k_\w+", r["Kernel_Name"])
    names[q][km.group(0) if km else r["Kernel_Name"][:30]] += e - s
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = None; hist = collections.defaultdict(float)
for t, d in ev:
    if last is not None and depth > 0: busy += t - last; hist[depth] += t - last
    depth += d; last = t
wall = max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])
print("wall %.2f ms | any kernel running %.2f ms | sum of durations %.2f ms" % (wall / 1e6, busy / 1e6, sum(per.values()) / 1e6))
print("time by number of kernels in flight:", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
for q, v in sorted(per.items(), key=lambda x: -x[1]):
    top = sorted(names[q].items(), key=lambda x: -x[1])[:4]
    print("queue", q, "%.2f ms" % (v / 1e6), [(n, round(t / 1e6, 2)) for n, t in top])

# launches of one kernel name by grid size (which of the step's launches of a shared kernel is the expensive one)
by = collections.defaultdict(list)
for r in rows:
    m = re.search(r"k_\w+", r["Kernel_Name"])
    nm = m.group(0) if m else r["Kernel_Name"][:28]
    if "filter" in nm or "resolve" in nm or "regression" in nm:
        by[(nm, r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (nm, g), v in sorted(by.items()):
    print("%-30s grid %-10s launches %4d  avg %7.1f us  min %7.1f  max %7.1f" % (nm, g, len(v), sum(v) / len(v), min(v), max(v)))
