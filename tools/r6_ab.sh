#!/bin/bash
# A/B on one box, alternating: this tree against the tree under _old/ (git worktree add _old <commit>, built) — the wide loop of 25 chains at
# configs[4]'s size, configs[3] (one chain, Hausdorff), the headline window.  usage: tools/r6_ab.sh [reps]
cd $GRAFT_REPO_ROOT
B="--cpu-steps 0 --profile-steps 0 --root-sampler-leg 0 --extra-configs= --many-chains 0 --dropin-leg 0"
leg() { # dir tag
  (cd $1
   w=$(ICP_HOST_DEVICE_LOOP=1 timeout 600 python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x_$2.npz 2>&1 | tail -1 | awk '{print $4, $5}')
   c3=$(python3 bench.py --config 3 --steps 600 --warmup 100 $B 2>/dev/null | python3 -c "import sys,json; print('%d' % json.loads(sys.stdin.read())['value'])")
   h=$(python3 bench.py --steps 20 --warmup 5 $B 2>/dev/null | python3 -c "import sys,json; print('%d' % json.loads(sys.stdin.read())['value'])")
   hl=$(python3 bench.py --steps 3000 --warmup 200 $B 2>/dev/null | python3 -c "import sys,json; print('%d' % json.loads(sys.stdin.read())['value'])")
   echo "$2: wide loop 25 x 200: $w | config3 $c3 | headline window $h | 3000 steps $hl")
}
for rep in $(seq 1 ${1:-2}); do leg . new; leg _old old; done
python3 - <<'PY'
import numpy as np
a, b = np.load('/tmp/x_new.npz'), np.load('/tmp/x_old.npz')
print('wide loop records identical:', all(np.array_equal(a[k], b[k]) for k in ('a', 'single', 'b')))
PY
