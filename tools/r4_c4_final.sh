#!/bin/bash
# configs[4] through bench.py at three chain lengths (whole job: contexts, chains, steps, records)
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
for steps in 50 300 3000; do
  echo "steps $steps"; timeout 900 python bench.py --config 4 --steps $steps --warmup 5 $B 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print(d['value'], d['config'])
print({k: d.get(k) for k in ('ms_per_step',)}, str(d.get('roofline'))[:600])
"
done
