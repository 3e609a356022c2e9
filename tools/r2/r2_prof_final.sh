# round-2 final profiles: kernel stats of the three configurations' bench runs (+ the 20-step window of the headline one)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
B="--many-chains 0 --cpu-steps 0 --profile-steps 0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1 -o s -- python3 bench.py --steps 2000 --warmup 200 $B > $O/c1.json 2> $O/c1.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1_20 -o s -- python3 bench.py --steps 20 --warmup 5 $B > $O/c1_20.json 2> $O/c1_20.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o s -- python3 bench.py --config 2 --steps 600 --warmup 50 $B > $O/c2.json 2> $O/c2.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o s -- python3 bench.py --config 3 --steps 600 --warmup 50 $B > $O/c3.json 2> $O/c3.err
for c in c1 c1_20 c2 c3; do
  f=$(find $O/$c -name '*kernel_stats.csv' | head -1)
  cp $f $O/${c}_kernel_stats.csv
  find $O/$c -name '*kernel_trace.csv' -delete
  grep -o '"value": [0-9.]*' $O/$c.json | head -1
done
