#!/bin/bash
# the 20-step window and the steady state with the direct route above several acceptance thresholds
B="--gpus 1 --many-chains 0 --cpu-steps 0 --profile-steps 0"
for thr in 2.0 0.6 0.4 0.0; do
  for k in 1 2 3; do
    echo -n "ICP_DIRECT_ABOVE=$thr 20-step: "; ICP_DIRECT_ABOVE=$thr timeout 120 python bench.py $B --steps 20 --warmup 5 2>/dev/null | grep -o '"value": [0-9.]*'
  done
  echo -n "ICP_DIRECT_ABOVE=$thr 3000-step: "; ICP_DIRECT_ABOVE=$thr timeout 200 python bench.py $B 2>/dev/null | grep -o '"value": [0-9.]*'
done
