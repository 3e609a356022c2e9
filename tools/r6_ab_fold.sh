#!/bin/bash
# A/B on one box: depth of the folded regression's operand pipeline (libraries built with -DICP_FOLD_DEPTH=1|2|3 beside the tree's 4) — the
# 25-chain wide loop at configs[4]'s size, alternating; records must be identical.  usage: tools/r6_ab_fold.sh [reps]
cd $GRAFT_REPO_ROOT
export ICP_WIDE_LOOP_TIMING=1 ICP_HOST_DEVICE_LOOP=1
P=$GRAFT_REPO_ROOT/icp-proposal_amd
for rep in $(seq 1 ${1:-2}); do
  for v in 4:libicp_proposal_amd_testhooks.so 1:libicp_fold_d1.so 2:libicp_fold_d2.so 3:libicp_fold_d3.so; do
    d=${v%%:*}; lib=${v##*:}
    [ -f $P/$lib ] || continue
    echo "depth $d: $(ICP_LIBRARY_PATH=$P/$lib timeout 600 python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x_d$d.npz 2>&1 | grep -oE '25 chains x 200 steps: enqueue [0-9.]+ ms, drain [0-9.]+ ms|mode 1 [0-9]+ it/s' | tr '\n' ' ')"
  done
done
python3 - <<'PY'
import numpy as np, os
a = np.load('/tmp/x_d1.npz')
for d in (2, 3, 4):
    f = '/tmp/x_d%d.npz' % d
    if os.path.exists(f):
        b = np.load(f)
        print('depth', d, 'records identical to depth 1:', all(np.array_equal(a[k], b[k]) for k in a.files))
PY
