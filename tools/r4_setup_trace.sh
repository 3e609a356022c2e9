#!/bin/bash
# HIP API time of a configs[4]-style job through bench.py --config 4 (10 targets x 10 chains x 50 steps): which runtime calls the
# set-up (contexts, chains, first steps, close) spends its time in
mkdir -p gpurun_out/setup_trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --hip-runtime-trace --stats -d gpurun_out/setup_trace -o s --output-format csv -- python3 bench.py --config 4 --steps 50 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0 > gpurun_out/setup_trace/run.json 2> gpurun_out/setup_trace/run.err
f=$(find gpurun_out/setup_trace -name "*hip_api_stats.csv" | head -1)
head -25 $f
find gpurun_out/setup_trace -name "*_trace.csv" -delete
grep -o '"value": [0-9.]*' gpurun_out/setup_trace/run.json | head -1
