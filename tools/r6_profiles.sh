#!/bin/bash
# round-6 profiles (run on the GPU box from the repository root; results under gpurun_out/r6prof, the summaries are then copied to
# profiles/r06_*).  Sections (first argument, default all):
#   stats   rocprofv3 --kernel-trace --stats of the driver's 20-step window, the 3,000-step chain, configs[2], [3], [4], femur-200, the
#           64-chain loop, the 25-chain wide loop (+ its queue overlap); HIP-event tables without device-side waits (bench.py --events-out)
#   pmc     counters, separate passes (tools/r6_pmc.py): FETCH_SIZE | WRITE_SIZE | eight SQ counters + GRBM_GUI_ACTIVE | TCC hit / miss, for
#           configs[1], [2], [3], the 64-chain loop AND the regime round 5 built: the 25-chain wide loop at configs[4]'s size and bench.py --config 4
#   pmcwide only the last two regimes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6prof; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0 --dropin-leg 0"
WHAT=${1:-all}
stats_of() { # name, title, cmd...
  n=$1; t=$2; shift; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- "$@" > $O/$n.json 2> $O/$n.err
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1)
  python3 tools/stats_md.py $f "$t" > $O/r06_${n}_kernel_stats.md
}
run() { # name, bench args...
  n=$1; shift
  stats_of $n "rocprofv3 --kernel-trace --stats -- python3 bench.py $* $B" python3 bench.py "$@" $B
  find $O/$n -name '*kernel_trace.csv' -delete
  echo "$n: $(grep -o '"value": [0-9.]*' $O/$n.json | head -1)"
}
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
run bench20 --steps 20 --warmup 5
run bench --steps 3000 --warmup 200
run config2 --config 2 --steps 400 --warmup 40
run config3 --config 3 --steps 600 --warmup 100
run config4 --config 4 --targets 2 --chains 10 --steps 300 --warmup 5
run dropin20 --steps 20 --warmup 5 --fused 3
export ICP_HOST_DEVICE_LOOP=1
stats_of device_loop64 "ICP_HOST_DEVICE_LOOP=1 rocprofv3 --kernel-trace --stats -- python3 tools/r3_device_loop.py 64 300 eigen" python3 tools/r3_device_loop.py 64 300 eigen /tmp/x.npy
find $O/device_loop64 -name '*kernel_trace.csv' -delete
stats_of wide_loop25 "ICP_HOST_DEVICE_LOOP=1 rocprofv3 --kernel-trace --stats -- python3 tools/r5_wide_loop.py facefull 25 200" python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x.npz
python3 tools/trace_overlap.py $(find $O/wide_loop25 -name '*kernel_trace.csv' | head -1) 0.5 > $O/r06_wide_loop25_queue_overlap.txt
find $O/wide_loop25 -name '*kernel_trace.csv' -delete; tail -1 $O/wide_loop25.json | cut -c1-160
stats_of femur200_loop3 "ICP_HOST_DEVICE_LOOP=1 rocprofv3 --kernel-trace --stats -- python3 tools/r5_wide_loop.py femur200 3 200" python3 tools/r5_wide_loop.py femur200 3 200 /tmp/x.npz
find $O/femur200_loop3 -name '*kernel_trace.csv' -delete; tail -1 $O/femur200_loop3.json | cut -c1-160
unset ICP_HOST_DEVICE_LOOP
# ---- wait-free per-kernel durations (HIP events on the launch streams, device-side waits taken out): un-profiled runs
E="--many-chains 0 --cpu-steps 0 --extra-configs= --root-sampler-leg 0 --dropin-leg 0"
python3 bench.py --steps 20 --warmup 5 --profile-steps 300 $E --events-out $O/r06_bench_event_durations.json > $O/ev_bench20.json 2> $O/ev_bench20.err
python3 bench.py --config 2 --steps 400 --warmup 40 --profile-steps 200 $E --events-out $O/r06_config2_event_durations.json > $O/ev_c2.json 2> $O/ev_c2.err
python3 bench.py --config 3 --steps 600 --warmup 100 --profile-steps 200 $E --events-out $O/r06_config3_event_durations.json > $O/ev_c3.json 2> $O/ev_c3.err
fi
# ---- PMC
SQC="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
TCC="TCC_HIT_sum TCC_MISS_sum"
pmc() { # regime, tag, counters (quoted), cmd...
  n=$1; tag=$2; c=$3; shift; shift; shift
  d=$O/pmc_${n}_$tag
  timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- "$@" > $d.log 2> $d.err
}
passes() { # regime, cmd...
  n=$1; shift
  pmc $n F FETCH_SIZE "$@"; pmc $n W WRITE_SIZE "$@"; pmc $n S "$SQC" "$@"; pmc $n L "$TCC" "$@"
}
cc() { find $O/pmc_$1_$2 -name '*counter_collection.csv' | head -1; }
spec() { echo "$1=$(cc $1 F),$(cc $1 W),$(cc $1 S),$(cc $1 L)"; }
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ] || [ "$WHAT" = pmcwide ]; then
export ICP_HOST_DEVICE_LOOP=1
passes wide_loop25 python3 tools/r5_wide_loop.py facefull 25 12 /tmp/x.npz
unset ICP_HOST_DEVICE_LOOP
passes config4 python3 bench.py --config 4 --targets 1 --chains 25 --steps 20 --warmup 2 $B
python3 tools/r6_pmc.py $O/r06_pmc $(spec wide_loop25) $(spec config4) > $O/pmc_collect_wide.log 2>&1; cat $O/pmc_collect_wide.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
export ICP_NO_PIPELINE=1
passes config1 python3 bench.py --steps 60 --warmup 10 $B
passes config2 python3 bench.py --config 2 --steps 40 --warmup 5 $B
passes config3 python3 bench.py --config 3 --steps 40 --warmup 5 $B
export ICP_HOST_DEVICE_LOOP=1
passes many_chains python3 tools/r3_device_loop.py 64 12 eigen /tmp/x.npy
unset ICP_HOST_DEVICE_LOOP
unset ICP_NO_PIPELINE
python3 tools/r6_pmc.py $O/r06_pmc $(spec config1) $(spec config2) $(spec config3) $(spec many_chains) > $O/pmc_collect.log 2>&1; cat $O/pmc_collect.log
fi
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
ls $O | head -80
