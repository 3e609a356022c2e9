#!/usr/bin/env python3
"""Per-kernel hardware counters of one regime from separate rocprofv3 --pmc passes (MI355X_MICROARCH.md "rocprofv3 PMC slots", §HBM):

  pass F   FETCH_SIZE                    (KiB; on gfx950 it tallies 64 B per 128-B request: the read side is doubled)
  pass W   WRITE_SIZE                    (KiB)
  pass S   SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES
           + GRBM_GUI_ACTIVE             (eight SQ slots and one GRBM slot: independent blocks, one pass)
  pass L   TCC_HIT_sum TCC_MISS_sum      (L2 hit rate)

usage: r6_pmc.py <out-prefix> <regime>=<F.csv>,<W.csv>,<S.csv>[,<L.csv>] ...
writes <out-prefix>_traffic.json (hbm_bytes_per_launch per kernel: what bench.py's `traffic` reads), <out-prefix>_mfma.json (MFMA-busy
fraction per launch of the kernels that issue matrix instructions: bench.py's `mfma` blocks) and <out-prefix>_sq.json (waves, average
waves per SIMD, VALU instructions, LDS bank-conflict share, share of wave cycles parked — per kernel, and per grid size for the
searches' filter launches); all three merge into existing files.  Medians over a kernel's launches; `median_us` from the S pass's
dispatch timestamps (a counter pass lets one kernel run at a time: durations WITHOUT neighbours).

Kernel keys: the function's own name (k_wide_regression_fold, k_tri_solve_many, k_wide_instance<8>, …); the filter launches also per grid
size (k_wide_filter@573440); plus the aliases rounds 3-5 filed launches under (every launch whose name contains `k_tridiag` under
k_tridiag, …), which bench.py's per-config lookups use."""
import collections, csv, json, os, re, statistics, sys

N_SIMD, N_XCD = 1024, 8
ALIASES = ["k_wide_instance", "k_wide_filter", "k_wide_resolve", "k_wide_regression", "k_wide_propose", "k_wide_prepare", "k_sum_partials",
           "k_mh_decide", "k_mh_front", "k_step_begin", "k_step_filter", "k_step_resolve", "k_step_regression", "k_step_finish", "k_posterior_eigen",
           "k_posterior_root", "k_tridiag", "k_tri_solve", "k_tri_gemm", "k_tri_back", "k_posterior_factor", "k_surface_filter", "k_surface_resolve",
           "k_vertex_filter", "k_transition_tails", "k_instance", "k_regression_mfma", "k_dist_stats", "k_propose"]
PER_GRID = ("k_wide_filter", "k_step_filter", "k_wide_resolve")
KEEP_TEMPLATE = ("k_wide_instance", "k_tridiag_many", "k_tri_solve_many", "k_tri_back_many", "k_mhw_decide")
MFMA_KERNELS = ("regression", "k_tri_gemm", "k_tri_back", "k_posterior_eigen")


def keys_of(row):
    name = row["Kernel_Name"]
    m = re.search(r"(k_[A-Za-z0-9_]+)(<[^>(]*>)?", name)
    if not m:
        return []
    base = m.group(1)
    exact = base + (m.group(2).replace(" ", "") if m.group(2) and base in KEEP_TEMPLATE else "")
    out = [exact]
    if exact != base:
        out.append(base)
    if base in PER_GRID and row.get("Grid_Size"):
        out.append("%s@%s" % (base, row["Grid_Size"]))
    for a in ALIASES:
        if a in name and a not in out:
            out.append(a)
            break
    return out


def read(path):
    """-> {key: {dispatch: {counter: value}}}, {key: {dispatch: us}}, {key: {dispatch: (grid, workgroup)}}"""
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    dur = collections.defaultdict(dict)
    shape = collections.defaultdict(dict)
    if not path or not os.path.exists(path):
        return per, dur, shape
    for r in csv.DictReader(open(path)):
        d = r.get("Dispatch_Id") or r.get("Correlation_Id")
        for k in keys_of(r):
            per[k][d][r["Counter_Name"]] = per[k][d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k][d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if r.get("Grid_Size"):
                shape[k][d] = (int(r["Grid_Size"]), int(r.get("Workgroup_Size") or 0))
    return per, dur, shape


def med(per, k, counter):
    v = [c[counter] for c in per.get(k, {}).values() if counter in c]
    return statistics.median(v) if v else None


def load(path):
    return json.load(open(path)) if os.path.exists(path) else {}


prefix = sys.argv[1]
traffic, mfma, sq = load(prefix + "_traffic.json"), load(prefix + "_mfma.json"), load(prefix + "_sq.json")
traffic["recipe"] = ("rocprofv3 --kernel-trace --pmc <COUNTER> -- <command>, one pass per counter (FETCH_SIZE, WRITE_SIZE); hbm_bytes_per_launch = "
                     "2*FETCH_SIZE*1024 + WRITE_SIZE*1024, medians over the launches (MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE counts 64 B per 128-B "
                     "request); host-stepped configurations with ICP_NO_PIPELINE=1 (a counter pass lets one kernel run at a time)")
mfma["recipe"] = ("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES … GRBM_GUI_ACTIVE -- <command> (tools/r6_profiles.sh); per kernel the median over its "
                  "launches; busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (%d x GRBM_GUI_ACTIVE / %d); median_us from the same pass's dispatch timestamps" % (N_SIMD, N_XCD))
sq["recipe"] = ("one rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY "
                "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (+ one with TCC_HIT_sum TCC_MISS_sum); medians per kernel over its launches.  avg_waves_per_simd = "
                "4 x SQ_WAVE_CYCLES (quad-cycles) / (%d SIMDs x GRBM_GUI_ACTIVE / %d XCDs); lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                "parked_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves waiting at s_waitcnt / barriers)" % (N_SIMD, N_XCD))
for spec in sys.argv[2:]:
    regime, files = spec.split("=")
    f = (files.split(",") + ["", "", "", ""])[:4]
    (pf, _, _), (pw, _, _), (ps, ds, shp), (pl, _, _) = read(f[0]), read(f[1]), read(f[2]), read(f[3])
    t_cfg, m_cfg, s_cfg = {}, {}, {}
    for k in sorted(set(pf) | set(ps)):
        fk, wk = med(pf, k, "FETCH_SIZE"), med(pw, k, "WRITE_SIZE")
        if fk is not None:
            t_cfg[k] = {"launches": len(pf[k]), "FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk or 0.0, "gfx950_fetch_correction": 2.0,
                        "hbm_bytes_per_launch": int(fk * 1024 * 2.0 + (wk or 0.0) * 1024)}
        if k not in ps:
            continue
        act, busy = med(ps, k, "GRBM_GUI_ACTIVE"), med(ps, k, "SQ_VALU_MFMA_BUSY_CYCLES")
        us = statistics.median(ds[k].values()) if ds.get(k) else None
        if busy is not None and act and any(t in k for t in MFMA_KERNELS):
            frac = [c["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * c["GRBM_GUI_ACTIVE"] / N_XCD) for c in ps[k].values()
                    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0]
            m_cfg[k] = {"SQ_VALU_MFMA_BUSY_CYCLES": {"launches": len(ps[k]), "median": busy}, "GRBM_GUI_ACTIVE": {"launches": len(ps[k]), "median": act},
                        "busy_frac_median": statistics.median(frac) if frac else None, "median_us": us}
        row = {"launches": len(ps[k]), "median_us": us}
        grids = sorted(set(shp.get(k, {}).values()))
        if grids:
            row["grid_threads"], row["workgroup"] = (grids[0][0] if len(grids) == 1 else [g[0] for g in grids]), grids[0][1]
        for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                  "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            v = med(ps, k, c)
            if v is not None:
                row[c] = v
        if act and row.get("SQ_WAVE_CYCLES") is not None:
            row["avg_waves_per_simd"] = 4.0 * row["SQ_WAVE_CYCLES"] / (N_SIMD * act / N_XCD)
        if row.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_share"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
        if row.get("SQ_WAVE_CYCLES"):
            row["parked_share"] = row.get("SQ_WAIT_ANY", 0.0) / row["SQ_WAVE_CYCLES"]
        if busy is not None and act:
            row["mfma_busy_frac"] = busy / (N_SIMD * act / N_XCD)
        hit, miss = med(pl, k, "TCC_HIT_sum"), med(pl, k, "TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            row["l2_hit_rate"] = hit / (hit + miss)
        if k in t_cfg:
            row["hbm_bytes_per_launch"] = t_cfg[k]["hbm_bytes_per_launch"]
        s_cfg[k] = row
    if t_cfg:
        traffic[regime] = t_cfg
    if m_cfg:
        mfma[regime] = m_cfg
    if s_cfg:
        sq[regime] = s_cfg
    print(regime, "kernels:", len(s_cfg), "| traffic rows", len(t_cfg), "| mfma rows", len(m_cfg))
for path, obj in ((prefix + "_traffic.json", traffic), (prefix + "_mfma.json", mfma), (prefix + "_sq.json", sq)):
    json.dump(obj, open(path, "w"), indent=1)
