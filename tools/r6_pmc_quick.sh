#!/bin/bash
# one SQ counter pass of the 25-chain wide loop -> per-kernel / per-grid table (developer aid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q; mkdir -p $O
SQC="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
export ICP_HOST_DEVICE_LOOP=1
timeout 1200 rocprofv3 --kernel-trace --pmc $SQC --output-format csv -d $O/S -o p -- python3 tools/r5_wide_loop.py facefull 25 12 /tmp/x.npz > $O/S.log 2> $O/S.err
f=$(find $O/S -name '*counter_collection.csv' | head -1)
rm -f $O/q_traffic.json $O/q_mfma.json $O/q_sq.json
python3 tools/r6_pmc.py $O/q wide_loop25=,,$f
python3 - <<'PY'
import json
sq=json.load(open('gpurun_out/r6q/q_sq.json'))['wide_loop25']
for k,v in sorted(sq.items()):
    if '@' in k:
        print("%-28s n=%4d us=%8.1f waves/simd=%5.2f parked=%4.2f valu=%12.0f waves=%8.0f" % (k, v['launches'], v.get('median_us') or 0, v.get('avg_waves_per_simd',0), v.get('parked_share',0), v.get('SQ_INSTS_VALU',0), v.get('SQ_WAVES',0)))
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
