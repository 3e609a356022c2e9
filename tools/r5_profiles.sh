#!/bin/bash
# round-5 profiles (run on the GPU box from the repository root; results under gpurun_out/r5prof, the summaries are then copied to
# profiles/r05_*):
#   * rocprofv3 --kernel-trace --stats of the driver's 20-step window, the 3,000-step chain, configs[2], [3], [4], the 64-chain loop
#   * the HIP-event table of the same windows WITHOUT device-side waits (bench.py --events-out)
#   * PMC: FETCH_SIZE / WRITE_SIZE in separate passes (HBM bytes per launch); SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in ONE pass
#     (SQ and GRBM slots are independent) -> MFMA-busy fraction per launch of the projection kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5prof; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
run() { # name, args...
  n=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 bench.py "$@" $B > $O/$n.json 2> $O/$n.err
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1)
  python3 tools/stats_md.py $f "rocprofv3 --kernel-trace --stats -- python3 bench.py $* $B" > $O/r05_${n}_kernel_stats.md
  find $O/$n -name '*kernel_trace.csv' -delete
  echo "$n: $(grep -o '"value": [0-9.]*' $O/$n.json | head -1)"
}
if [ "${1:-all}" != "pmc" ]; then
run bench20 --steps 20 --warmup 5
run bench --steps 3000 --warmup 200
run config2 --config 2 --steps 400 --warmup 40
run config3 --config 3 --steps 600 --warmup 100
run config4 --config 4 --targets 2 --chains 10 --steps 300 --warmup 5
n=device_loop64
ICP_HOST_DEVICE_LOOP=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 tools/r3_device_loop.py 64 300 eigen /tmp/x.npy > $O/$n.log 2> $O/$n.err
f=$(find $O/$n -name '*kernel_stats.csv' | head -1)
python3 tools/stats_md.py $f "ICP_HOST_DEVICE_LOOP=1 rocprofv3 --kernel-trace --stats -- python3 tools/r3_device_loop.py 64 300 eigen" > $O/r05_${n}_kernel_stats.md
find $O/$n -name '*kernel_trace.csv' -delete; tail -1 $O/$n.log | cut -c1-160
# ---- the wide step's on-device loop: 25 chains of the configs[4] size (N = 28,561, rank 200), 200 steps
n=wide_loop25
ICP_HOST_DEVICE_LOOP=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x.npz > $O/$n.log 2> $O/$n.err
f=$(find $O/$n -name '*kernel_stats.csv' | head -1)
python3 tools/stats_md.py $f "ICP_HOST_DEVICE_LOOP=1 rocprofv3 --kernel-trace --stats -- python3 tools/r5_wide_loop.py facefull 25 200" > $O/r05_${n}_kernel_stats.md
python3 tools/trace_overlap.py $(find $O/$n -name '*kernel_trace.csv' | head -1) 0.5 > $O/r05_${n}_queue_overlap.txt
find $O/$n -name '*kernel_trace.csv' -delete; tail -1 $O/$n.log | cut -c1-160
# ---- wait-free per-kernel durations (HIP events on the launch streams, device-side waits taken out): un-profiled runs
E="--many-chains 0 --cpu-steps 0 --extra-configs= --root-sampler-leg 0"
python3 bench.py --steps 20 --warmup 5 --profile-steps 300 $E --events-out $O/r05_bench_event_durations.json > $O/ev_bench20.json 2> $O/ev_bench20.err
python3 bench.py --config 2 --steps 400 --warmup 40 --profile-steps 200 $E --events-out $O/r05_config2_event_durations.json > $O/ev_c2.json 2> $O/ev_c2.err
python3 bench.py --config 3 --steps 600 --warmup 100 --profile-steps 200 $E --events-out $O/r05_config3_event_durations.json > $O/ev_c3.json 2> $O/ev_c3.err
fi
# ---- PMC
pmc() { # name, counters (quoted), cmd...
  n=$1; c=$2; shift; shift
  d=$O/pmc_${n}_$(echo $c | tr ' ' '+')
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- "$@" > $d.log 2> $d.err
}
export ICP_NO_PIPELINE=1
for c in FETCH_SIZE WRITE_SIZE; do
  pmc c1 $c python3 bench.py --steps 60 --warmup 10 $B
  pmc c2 $c python3 bench.py --config 2 --steps 40 --warmup 5 $B
  pmc c3 $c python3 bench.py --config 3 --steps 40 --warmup 5 $B
  ICP_HOST_DEVICE_LOOP=1 pmc mc $c python3 tools/r3_device_loop.py 64 12 eigen /tmp/x.npy
done
M="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
pmc c1 "$M" python3 bench.py --steps 60 --warmup 10 $B
pmc c2 "$M" python3 bench.py --config 2 --steps 40 --warmup 5 $B
pmc c3 "$M" python3 bench.py --config 3 --steps 40 --warmup 5 $B
ICP_HOST_DEVICE_LOOP=1 pmc mc "$M" python3 tools/r3_device_loop.py 64 12 eigen /tmp/x.npy
unset ICP_NO_PIPELINE
cc() { find $O/pmc_$1_$2 -name '*counter_collection.csv' | head -1; }
python3 tools/pmc_collect.py $O/r05_pmc_traffic.json config1=$(cc c1 FETCH_SIZE),$(cc c1 WRITE_SIZE) config2=$(cc c2 FETCH_SIZE),$(cc c2 WRITE_SIZE) \
  config3=$(cc c3 FETCH_SIZE),$(cc c3 WRITE_SIZE) many_chains=$(cc mc FETCH_SIZE),$(cc mc WRITE_SIZE) > $O/pmc_collect.log 2>&1
python3 tools/pmc_mfma.py $O/r05_pmc_mfma.json config1=$(cc c1 SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) config2=$(cc c2 SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) \
  config3=$(cc c3 SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) many_chains=$(cc mc SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) > $O/pmc_mfma.log 2>&1
head -c 1500 $O/pmc_mfma.log
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
ls $O | head -60
