#!/bin/bash
# A/B (this tree / _old), each run under its own timeout: the 25-chain wide loop only
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${1:-2}); do
  for d in . _old; do
    (cd $d; echo "$d: $(ICP_HOST_DEVICE_LOOP=1 timeout 120 python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x_$(basename $d).npz 2>&1 | tail -1 | awk '{print $4, $5, $NF}' | cut -c1-80)")
  done
done
