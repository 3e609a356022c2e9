#!/bin/bash
# kernel trace of configs[4] with 30 chains of 3 targets per submission: timeline of a few steps near the end
mkdir -p gpurun_out/tr4
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tr4 -o c4 --output-format csv -- python3 tools/r4_c4_many.py ${1:-3} > gpurun_out/tr4/run.log 2>&1
f=$(find gpurun_out/tr4 -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
python3 tools/timeline.py $f $((n * 8 / 10)) 160 > gpurun_out/tr4/timeline.txt
find gpurun_out/tr4 -name "*kernel_trace.csv" -delete
grep "targets" gpurun_out/tr4/run.log
