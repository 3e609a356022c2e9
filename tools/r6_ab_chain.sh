#!/bin/bash
# A/B on one box: the critical chain of the on-device wide loop on ONE queue (ICP_WIDE_LOOP_CHAIN_MAIN=0: the older layout) —
# 25 chains x 200 steps at configs[4]'s size, alternating; the records of both must be identical.  usage: tools/r6_ab_chain.sh [reps]
cd $GRAFT_REPO_ROOT
export ICP_LIBRARY_PATH=$GRAFT_REPO_ROOT/icp-proposal_amd/libicp_proposal_amd_testhooks.so ICP_WIDE_LOOP_TIMING=1 ICP_HOST_DEVICE_LOOP=1
for rep in $(seq 1 ${1:-3}); do
  echo "on : $(timeout 600 python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x_on.npz 2>&1 | grep -E 'wide loop|it/s' | tr '\n' ' ')"
  echo "off: $(ICP_WIDE_LOOP_CHAIN_MAIN=0 timeout 600 python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x_off.npz 2>&1 | grep -E 'wide loop|it/s' | tr '\n' ' ')"
done
python3 - <<'PY'
import numpy as np
a, b = np.load('/tmp/x_on.npz'), np.load('/tmp/x_off.npz')
print('records identical:', all(np.array_equal(a[k], b[k]) for k in a.files))
PY
