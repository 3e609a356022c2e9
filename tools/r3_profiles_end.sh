#!/bin/bash
# end of round 3: kernel statistics of the bench legs (the PMC traffic passes of tools/r3_profiles.sh are not repeated: those kernels did not change)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
run() { # name, args...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 bench.py "$@" $B > $O/$n.json 2> $O/$n.err
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  find $O/$n -name '*kernel_trace.csv' -delete
  echo "$n: $(grep -o '"value": [0-9.]*' $O/$n.json | head -1)"
}
run c1 --steps 3000 --warmup 200
run c1_20 --steps 20 --warmup 5
run c2 --config 2 --steps 600 --warmup 50
run c3 --config 3 --steps 600 --warmup 50
run c3root --config 3 --steps 600 --warmup 50 --sampler cholesky-root
