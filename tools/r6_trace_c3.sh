#!/bin/bash
# a window of configs[3] (one chain, rank 200, full-mesh Hausdorff; the wide step from the host) as a timeline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6c3; mkdir -p $O
B="--cpu-steps 0 --profile-steps 0 --root-sampler-leg 0 --extra-configs= --many-chains 0 --dropin-leg 0"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python3 bench.py --config 3 --steps ${STEPS:-200} --warmup 50 $B > $O/bench.log 2>&1
f=$(find $O/t -name '*kernel_trace.csv' | head -1)
python3 tools/r6_trace_window.py $f ${AT:-0.6} ${N:-90} > $O/window.txt
find $O/t -name '*kernel_trace.csv' -delete
tail -1 $O/bench.log | cut -c1-120
