#!/bin/bash
# round-3 profiles of configs[3] after the register-tiled factorisation: kernel stats (KL-basis sampler and Cholesky-root sampler),
# PMC traffic passes, merged into profiles/r03_pmc_traffic.json by tools/pmc_collect.py (config3 entry only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
run() { # name, args...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 bench.py "$@" $B > $O/$n.json 2> $O/$n.err
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1); cp $f $O/${n}_kernel_stats.csv
  find $O/$n -name '*kernel_trace.csv' -delete
  echo "$n: $(grep -o '"value": [0-9.]*' $O/$n.json | head -1)"
}
run c3 --config 3 --steps 600 --warmup 50
run c3root --config 3 --steps 600 --warmup 50 --sampler cholesky-root
export ICP_NO_PIPELINE=1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc3_$c -o p -- python3 bench.py --config 3 --steps 60 --warmup 10 $B > $O/pmc3_$c.json 2> $O/pmc3_$c.err
done
unset ICP_NO_PIPELINE
f=$(find $O/pmc3_FETCH_SIZE -name '*counter_collection.csv' | head -1); w=$(find $O/pmc3_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_collect.py $O/r03_pmc_traffic_c3.json config3=$f,$w
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
ls $O
