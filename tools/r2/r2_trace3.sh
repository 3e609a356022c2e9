#!/bin/bash
# kernel trace of the configs[3] chain: timeline of a few steps in the middle
mkdir -p gpurun_out/tr3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/tr3 -o c3 --output-format csv -- python3 bench.py --gpus 1 --config 3 --steps 80 --warmup 20 --cpu-steps 0 --many-chains 0 --profile-steps 0 > gpurun_out/tr3/bench.json 2> gpurun_out/tr3/bench.err
f=$(find gpurun_out/tr3 -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 2500 260 > gpurun_out/tr3/timeline.txt
find gpurun_out/tr3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/tr3/kernel_stats.csv
find gpurun_out/tr3 -name "*kernel_trace.csv" -delete
tail -c 300 gpurun_out/tr3/bench.json
