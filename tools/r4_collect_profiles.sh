#!/bin/bash
# gpurun_out/r4prof (tools/r4_profiles.sh, scratch) -> profiles/r04_* (tracked)
O=gpurun_out/r4prof
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
md() { python3 tools/stats_md.py $O/$1_kernel_stats.csv "$2" > profiles/r04_$1_kernel_stats.md; }
md bench20 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 $B   (the driver's window)"
md bench "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3000 --warmup 200 $B"
md config2 "rocprofv3 --kernel-trace --stats -- python3 bench.py --config 2 --steps 400 --warmup 40 $B"
md config3 "rocprofv3 --kernel-trace --stats -- python3 bench.py --config 3 --steps 600 --warmup 100 $B   (the wide step)"
md config4 "rocprofv3 --kernel-trace --stats -- python3 bench.py --config 4 --targets 2 --chains 10 --steps 300 --warmup 5 $B   (the wide step, chains side by side)"
md device_loop64 "ICP_HOST_DEVICE_LOOP=1 rocprofv3 --kernel-trace --stats -- python3 tools/r3_device_loop.py 64 300 eigen"
cp $O/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
cp $O/r04_pmc_mfma.json profiles/r04_pmc_mfma.json
ls -la profiles | grep r04
