#!/bin/bash
# macro-tile regression: parity tests, then rates of every configuration
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_face.py tests/test_gpu_parity.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -3
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done
echo c2; timeout 300 python bench.py --config 2 --steps 400 --warmup 40 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
echo c3; timeout 300 python bench.py --config 3 --steps 600 --warmup 50 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
timeout 150 python tools/r4_c4_many.py 3 2>&1 | grep "targets 3"
timeout 150 python tools/r4_c4_setup.py 10 2>&1 | grep "target [12]"
timeout 300 bash tools/r4_trace_c4.sh 3
head -45 gpurun_out/tr4/timeline.txt
