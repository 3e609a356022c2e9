#!/usr/bin/env python3
"""Turn raw rocprofv3 CSV output (gpurun_out/<run>/...) into the small summaries committed under profiles/.

usage: summarize.py stats <kernel_stats.csv> <out.md> "<command>"
       summarize.py pmc <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <kernel-substring>
PMC recipe (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are collected in SEPARATE --pmc passes, both in KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request, so the read side is doubled; WRITE_SIZE is exact for streaming stores.
"""
import collections
import csv
import json
import statistics
import sys


def stats(path, out, cmd):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        f.write("| kernel | calls | avg us | min us | max us | total ms | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            f.write("| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.1f |\n" % (
                r["Name"][:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
        f.write(f"\ntotal kernel time: {tot / 1e6:.2f} ms\n")


def pmc(fetch, write, out, kernel):
    def collect(path):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return acc
    fa, wa = collect(fetch), collect(write)
    res = {}
    for name in fa:
        if kernel in name:
            fk = statistics.median(fa[name])
            wk = statistics.median(wa.get(name, [0.0]))
            res = {"kernel": name[:120], "launches": len(fa[name]), "FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk,
                   "gfx950_fetch_correction": 2.0, "hbm_bytes_per_launch": int(fk * 1024 * 2.0 + wk * 1024),
                   "recipe": "separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md §HBM)"}
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:5])
    else:
        pmc(*sys.argv[2:6])
