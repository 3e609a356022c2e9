#!/bin/bash
# configs[4], 30 chains of 3 targets per submission: one lockstep group of 30 against two of 15 and three of 10
for g in 32 16 10 8; do
  echo "ICP_WIDE_GROUP=$g"; ICP_WIDE_GROUP=$g python tools/r4_c4_many.py 3 2>&1 | grep "targets 3"
done
echo "timing, ICP_WIDE_GROUP=16"; ICP_HOST_TIMING=1 ICP_WIDE_GROUP=16 python tools/r4_c4_many.py 3 2>&1 | grep "batch timing" | tail -3
