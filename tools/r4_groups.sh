#!/bin/bash
# configs[4] steady state against the number of lockstep groups (10 chains of one target)
for g in 1 2 3 4; do
  echo "groups $g"; ICP_LOCKSTEP_GROUPS=$g python tools/r4_c4_setup.py 10 2>&1 | grep "target [12]"
done
echo "timing, 2 groups"; ICP_HOST_TIMING=1 ICP_LOCKSTEP_GROUPS=2 python tools/r4_c4_setup.py 10 2>&1 | grep "batch timing" | tail -3
echo "timing, 4 groups"; ICP_HOST_TIMING=1 ICP_LOCKSTEP_GROUPS=4 python tools/r4_c4_setup.py 10 2>&1 | grep "batch timing" | tail -3
