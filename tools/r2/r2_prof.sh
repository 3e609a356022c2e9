# round-2 profiles: kernel stats of the headline run + PMC passes (separate runs, ICP_NO_PIPELINE=1: a counter pass serialises kernels)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c1 -o s -- python3 bench.py --steps 2000 --warmup 200 $B > $O/stats_c1.json 2> $O/stats_c1.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c1_20 -o s -- python3 bench.py --steps 20 --warmup 5 $B > $O/stats_c1_20.json 2> $O/stats_c1_20.err
export ICP_NO_PIPELINE=1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 200 --warmup 20 $B > $O/pmc_$c.json 2> $O/pmc_$c.err
done
find $O -name '*.csv' | xargs ls -la | awk '{print $5, $9}'
