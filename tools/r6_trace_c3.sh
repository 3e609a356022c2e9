#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c3; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0 --dropin-leg 0"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python3 bench.py --config ${CFG:-3} --steps 400 --warmup 100 $B > $O/t.json 2> $O/t.err
python3 tools/r6_trace_window.py $(find $O/t -name '*kernel_trace.csv' | head -1) 0.6 ${N:-90} > $O/window.txt
find $O/t -name '*kernel_trace.csv' -delete
cat $O/window.txt
