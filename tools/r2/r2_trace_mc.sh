#!/bin/bash
# kernel trace of the 64-chain batched leg
mkdir -p gpurun_out/trmc
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/trmc -o mc --output-format csv -- python3 bench.py --gpus 1 --steps 50 --warmup 5 --cpu-steps 0 --many-chains 64 --profile-steps 0 > gpurun_out/trmc/bench.json 2> gpurun_out/trmc/bench.err
f=$(find gpurun_out/trmc -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
python3 tools/timeline.py $f $(( n - 1200 )) 240 > gpurun_out/trmc/timeline.txt
find gpurun_out/trmc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/trmc/kernel_stats.csv
find gpurun_out/trmc -name "*kernel_trace.csv" -delete
tail -c 200 gpurun_out/trmc/bench.json
