#!/bin/bash
# the stream / pinned-block pools: configs[4] with short chains (whole job) with and without, then the GPU tests
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
for v in 1 0; do
  for steps in 50 300; do
    if [ $v = 1 ]; then export ICP_NO_POOL=1; else unset ICP_NO_POOL; fi
    echo "ICP_NO_POOL=${ICP_NO_POOL:-unset} steps $steps: $(timeout 600 python bench.py --config 4 --steps $steps --warmup 5 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
  done
done
unset ICP_NO_POOL
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done
