#!/usr/bin/env python3
"""MFMA-busy fraction per launch of the projection kernels from ONE rocprofv3 --pmc pass that collects SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE together (SQ and GRBM counter slots are independent: MI355X_MICROARCH.md "rocprofv3 PMC slots").

  busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)

SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_f64_16x16x4), summed over the chip; rocprofv3 reports GRBM_GUI_ACTIVE
summed over the 8 XCDs, so a launch lasted GRBM_GUI_ACTIVE / 8 cycles (the guide's "DVFS give-back" paragraph; the quotient reads high on
dispatches shorter than about 0.3 ms — these are 5-50 µs —, which makes busy_frac a LOWER bound here).

usage: pmc_mfma.py <out.json> <config-name>=<counter_collection.csv> ...   (merges into out.json if it exists)"""
import collections, csv, json, os, statistics, sys

KERNELS = ["k_step_regression", "k_wide_regression", "k_regression_mfma", "k_tri_gemm", "k_tri_back", "k_posterior_eigen"]
N_SIMD, N_XCD = 1024, 8


def short(name):
    for k in KERNELS:
        if k in name:
            return "k_step_regression" if k == "k_wide_regression" else k  # (the wide step's regression is timed under the same id)
    return None


out_path = sys.argv[1]
res = json.load(open(out_path)) if os.path.exists(out_path) else {}
res["recipe"] = ("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --config N … (ICP_NO_PIPELINE=1: a counter "
                 "pass lets one kernel run at a time); per kernel the median over its launches; busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (%d x "
                 "GRBM_GUI_ACTIVE / %d); median_us from the same pass's dispatch timestamps" % (N_SIMD, N_XCD))
for spec in sys.argv[2:]:
    name, path = spec.split("=")
    if not path or not os.path.exists(path):
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch -> counter -> value
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        d = r.get("Dispatch_Id") or r.get("Correlation_Id")
        per[k][d][r["Counter_Name"]] = per[k][d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k][d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cfg = {}
    for k, disp in per.items():
        busy = [v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in disp.values() if "SQ_VALU_MFMA_BUSY_CYCLES" in v]
        act = [v["GRBM_GUI_ACTIVE"] for v in disp.values() if "GRBM_GUI_ACTIVE" in v]
        frac = [v["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * v["GRBM_GUI_ACTIVE"] / N_XCD) for v in disp.values()
                if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0]
        if not busy:
            continue
        cfg[k] = {"SQ_VALU_MFMA_BUSY_CYCLES": {"launches": len(busy), "median": statistics.median(busy)},
                  "GRBM_GUI_ACTIVE": {"launches": len(act), "median": statistics.median(act) if act else None},
                  "busy_frac_median": statistics.median(frac) if frac else None,
                  "median_us": statistics.median(dur[k].values()) if dur[k] else None}
    res[name] = cfg
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: {kk: vv.get("busy_frac_median") for kk, vv in v.items()} for k, v in res.items() if k != "recipe"}))
