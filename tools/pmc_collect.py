#!/usr/bin/env python3
"""Per-kernel HBM traffic from separate rocprofv3 --pmc passes (MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE in SEPARATE
passes, both in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request -> the read side is doubled).
usage: pmc_collect.py <out.json> <config-name>=<fetch.csv>,<write.csv> ...   (merges into out.json if it exists)"""
import collections, csv, json, os, statistics, sys

SHORT = ["k_wide_instance", "k_wide_filter", "k_wide_resolve", "k_wide_regression", "k_wide_propose", "k_wide_prepare", "k_sum_partials",
         "k_mh_decide", "k_mh_front", "k_step_begin", "k_step_filter", "k_step_resolve", "k_step_regression", "k_step_finish", "k_posterior_eigen", "k_posterior_root",
         "k_tridiag", "k_tri_solve", "k_tri_gemm", "k_posterior_factor", "k_surface_filter", "k_surface_resolve", "k_vertex_filter",
         "k_transition_tails", "k_instance", "k_regression_mfma", "k_dist_stats", "k_propose"]


def short(name):
    for k in SHORT:
        if k in name:
            return k
    return None


def collect(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            acc[k].append(float(r["Counter_Value"]))
    return acc


out_path = sys.argv[1]
res = json.load(open(out_path)) if os.path.exists(out_path) else {}
res["recipe"] = ("rocprofv3 --kernel-trace --pmc <COUNTER> -- python3 bench.py --config N ..., one pass per counter (FETCH_SIZE, WRITE_SIZE), ICP_NO_PIPELINE=1 (a counter "
                 "pass lets one kernel run at a time: steps run on one stream, same kernels and grids); hbm_bytes_per_launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, medians "
                 "over the launches (MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE counts 64 B per 128-B request)")
for spec in sys.argv[2:]:
    name, files = spec.split("=")
    f, w = files.split(",")
    fa, wa = collect(f), collect(w)
    cfg = {}
    for k in fa:
        fk, wk = statistics.median(fa[k]), statistics.median(wa.get(k, [0.0]))
        cfg[k] = {"launches": len(fa[k]), "FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk, "gfx950_fetch_correction": 2.0,
                  "hbm_bytes_per_launch": int(fk * 1024 * 2.0 + wk * 1024)}
    res[name] = cfg
json.dump(res, open(out_path, "w"), indent=1)
print({k: {kk: vv["hbm_bytes_per_launch"] for kk, vv in v.items()} for k, v in res.items() if k != "recipe"})
