#!/usr/bin/env python3
"""overlap of the hardware queues in a rocprofv3 kernel trace: per queue the busy time, the union over queues, the wall span
usage: trace_overlap.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections
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
    names[q][r["Kernel_Name"].split("(")[0][-40:]] += e - s
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
