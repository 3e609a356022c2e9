#!/bin/bash
# block_matvec with sixteen entries in flight: parity tests, rates, kernel times of the tails / proposal launches
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_face.py tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done
echo c2; timeout 300 python bench.py --config 2 --steps 400 --warmup 40 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
echo c3; for i in 1 2; do timeout 300 python bench.py --config 3 --steps 600 --warmup 50 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done
timeout 150 python tools/r4_c4_many.py 3 2>&1 | grep "targets 3"
bash tools/r4_trace_c3.sh > /dev/null 2>&1; grep "k_transition_tails\|k_wide_propose\|k_posterior_factor" gpurun_out/tr3/timeline.txt | head -9
