#!/bin/bash
# memory-side counters of the on-device loop's batched launches (64 chains): HBM bytes and L2 hits / misses per launch
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcdl2; mkdir -p $O
export ICP_HOST_DEVICE_LOOP=1
i=0
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 tools/r3_device_loop.py 64 40 eigen /tmp/x.npy > $O/run$i.log 2>&1
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("icp::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in sorted(acc.items()):
    if "batch" in k or "mh_" in k or "eigen_rr" in k:
        print("%-44s " % k + "  ".join("%s/launch %.4g" % (cn, v / max(1, n[k][cn])) for cn, v in c.items()))
PY
  find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
done
