#!/bin/bash
# a window of the headline chain (configs[1], icp_chain_step) as a timeline — which launch ends when, on which queue
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6head; mkdir -p $O
B="--cpu-steps 0 --profile-steps 0 --root-sampler-leg 0 --extra-configs= --many-chains 0 --dropin-leg 0"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python3 bench.py --steps ${STEPS:-60} --warmup 10 $B > $O/bench.log 2>&1
f=$(find $O/t -name '*kernel_trace.csv' | head -1)
python3 tools/r6_trace_window.py $f ${AT:-0.5} ${N:-40} > $O/window.txt
find $O/t -name '*kernel_trace.csv' -delete
tail -1 $O/bench.log | cut -c1-200
cat $O/window.txt
