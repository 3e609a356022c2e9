#!/bin/bash
# round-4 profiles: kernel stats (rocprofv3 --kernel-trace --stats) of the driver's 20-step window, the 3,000-step chain, configs[2],
# configs[3] (wide step), configs[4] (wide step, 10 chains per target), the 64-chain on-device loop; PMC passes (FETCH_SIZE,
# WRITE_SIZE, separate; SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES) merged into profiles-style json by tools/pmc_collect.py.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4prof; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
run() { # name, args...
  n=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 bench.py "$@" $B > $O/$n.json 2> $O/$n.err
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1); cp $f $O/${n}_kernel_stats.csv
  find $O/$n -name '*kernel_trace.csv' -delete
  echo "$n: $(grep -o '"value": [0-9.]*' $O/$n.json | head -1)"
}
run bench20 --steps 20 --warmup 5
run bench --steps 3000 --warmup 200
run config2 --config 2 --steps 400 --warmup 40
run config3 --config 3 --steps 600 --warmup 100
run config4 --config 4 --targets 2 --chains 10 --steps 300 --warmup 5
n=device_loop64
ICP_HOST_DEVICE_LOOP=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 tools/r3_device_loop.py 64 300 eigen /tmp/x.npy > $O/$n.log 2> $O/$n.err
f=$(find $O/$n -name '*kernel_stats.csv' | head -1); cp $f $O/${n}_kernel_stats.csv; find $O/$n -name '*kernel_trace.csv' -delete; tail -1 $O/$n.log | cut -c1-160
# ---- PMC: HBM bytes per launch (separate passes per counter)
pmc() { # name, counter, cmd...
  n=$1; c=$2; shift; shift
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${n}_$c -o p -- "$@" > $O/pmc_${n}_$c.log 2> $O/pmc_${n}_$c.err
}
export ICP_NO_PIPELINE=1
for c in FETCH_SIZE WRITE_SIZE; do
  pmc c1 $c python3 bench.py --steps 60 --warmup 10 $B
  pmc c2 $c python3 bench.py --config 2 --steps 40 --warmup 5 $B
  pmc c3 $c python3 bench.py --config 3 --steps 40 --warmup 5 $B
  ICP_HOST_DEVICE_LOOP=1 pmc mc $c python3 tools/r3_device_loop.py 64 12 eigen /tmp/x.npy
done
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  pmc c2 $c python3 bench.py --config 2 --steps 40 --warmup 5 $B
  pmc c3 $c python3 bench.py --config 3 --steps 40 --warmup 5 $B
done
unset ICP_NO_PIPELINE
cc() { find $O/pmc_$1_$2 -name '*counter_collection.csv' | head -1; }
python3 tools/pmc_collect.py $O/r04_pmc_traffic.json config1=$(cc c1 FETCH_SIZE),$(cc c1 WRITE_SIZE) config2=$(cc c2 FETCH_SIZE),$(cc c2 WRITE_SIZE) \
  config3=$(cc c3 FETCH_SIZE),$(cc c3 WRITE_SIZE) many_chains=$(cc mc FETCH_SIZE),$(cc mc WRITE_SIZE) > $O/pmc_collect.log 2>&1
python3 - $O <<'PY'
import csv, sys, collections, glob, json, os
O = sys.argv[1]
res = {}
for cfg in ("c2", "c3"):
    out = {}
    for ctr in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
        fs = glob.glob(os.path.join(O, "pmc_%s_%s" % (cfg, ctr), "**", "*counter_collection.csv"), recursive=True)
        if not fs: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            name = r["Kernel_Name"]
            for k in ("k_step_regression", "k_wide_regression", "k_tri_gemm", "k_regression_mfma"):
                if k in name: acc[k].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {})[ctr] = {"launches": len(v), "median": sorted(v)[len(v) // 2]}
    res["config2" if cfg == "c2" else "config3"] = out
json.dump(res, open(os.path.join(O, "r04_pmc_mfma.json"), "w"), indent=1)
print(json.dumps(res)[:1500])
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
ls $O | head -50
