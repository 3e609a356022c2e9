#!/bin/bash
# round-4 profiles of the wide step: kernel stats of configs[3] (one chain) and configs[4] (10 chains of a target side by side)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
run() { # name, args...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 bench.py "$@" $B > $O/$n.json 2> $O/$n.err
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1); cp $f $O/${n}_kernel_stats.csv
  find $O/$n -name '*kernel_trace.csv' -delete
  echo "$n: $(grep -o '"value": [0-9.]*' $O/$n.json | head -1)"
}
run c3 --config 3 --steps 600 --warmup 50
run c4 --config 4 --targets 2 --chains 10 --steps 300 --warmup 5
ls $O
