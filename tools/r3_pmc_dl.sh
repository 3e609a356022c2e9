#!/bin/bash
# SQ counters of the on-device loop's batched launches (64 chains): where the waves of the chip-wide kernels spend their cycles
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcdl; mkdir -p $O
export ICP_HOST_DEVICE_LOOP=1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -o p -- python3 tools/r3_device_loop.py 64 40 eigen /tmp/x.npy > $O/run.log 2>&1
tail -2 $O/run.log | cut -c1-200
f=$(find $O/sq -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("icp::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:10]:
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print("%-42s launches %5d waves/launch %8.0f  wave-cycles/launch %.3g | parked %.2f active %.2f issue-stall %.2f" % (
        k, n[k], c["SQ_WAVES"] / max(n[k], 1), wc / max(n[k], 1), c["SQ_WAIT_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc))
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
