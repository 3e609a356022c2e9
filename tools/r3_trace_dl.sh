#!/bin/bash
# kernel trace of the on-device MH loop: timeline of a few steps in the middle + per-kernel stats
mkdir -p gpurun_out/dl
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/dl -o dl --output-format csv -- python3 tools/r3_device_loop.py ${2:-64} 120 ${1:-eigen} > gpurun_out/dl/run.log 2>&1
f=$(find gpurun_out/dl -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
python3 tools/timeline.py $f $((n * 3 / 4)) 90 > gpurun_out/dl/timeline.txt
find gpurun_out/dl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/dl/kernel_stats.csv
find gpurun_out/dl -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/dl/run.log
head -40 gpurun_out/dl/kernel_stats.csv | cut -c1-150
cat gpurun_out/dl/timeline.txt
