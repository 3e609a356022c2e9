#!/bin/bash
# round-3 profiles: kernel stats of the benches + PMC traffic passes for configs 1-3
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
run c1 --steps 3000 --warmup 200
run c1_20 --steps 20 --warmup 5
run c2 --config 2 --steps 600 --warmup 50
run c3 --config 3 --steps 600 --warmup 50
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mc64 -o s -- python3 bench.py --steps 50 --warmup 5 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0 --many-chains 64 > $O/mc64.json 2> $O/mc64.err
cp $(find $O/mc64 -name '*kernel_stats.csv' | head -1) $O/mc64_kernel_stats.csv; find $O/mc64 -name '*kernel_trace.csv' -delete
python3 -c "import json; d=json.load(open('$O/mc64.json')); print('mc64', d.get('many_chains'))"
export ICP_NO_PIPELINE=1
for cfg in 1 2 3; do
  st=200; [ $cfg = 3 ] && st=60
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc${cfg}_$c -o p -- python3 bench.py --config $cfg --steps $st --warmup 10 $B > $O/pmc${cfg}_$c.json 2> $O/pmc${cfg}_$c.err
  done
done
unset ICP_NO_PIPELINE
args=""
for cfg in 1 2 3; do
  f=$(find $O/pmc${cfg}_FETCH_SIZE -name '*counter_collection.csv' | head -1); w=$(find $O/pmc${cfg}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  args="$args config$cfg=$f,$w"
done
python3 tools/pmc_collect.py $O/r03_pmc_traffic.json $args
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
ls $O
