#!/bin/bash
# host-side API time of the configs[3] chain: rocprofv3 --hip-runtime-trace --stats
mkdir -p gpurun_out/ht3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --hip-runtime-trace --stats -d gpurun_out/ht3 -o c3 --output-format csv -- python3 /tmp/c3.py ${1:-cholesky-root} > gpurun_out/ht3/run.log 2>&1
grep rate gpurun_out/ht3/run.log
f=$(find gpurun_out/ht3 -name "*hip_api_stats.csv" | head -1)
head -25 $f | cut -c1-160
find gpurun_out/ht3 -name "*trace.csv" -delete
