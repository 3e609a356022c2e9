#!/bin/bash
# rate of the wide step's on-device loop by group count (test-hooks build: ICP_WIDE_LOOP_GROUPS), against host stepping
export ICP_LIBRARY_PATH=$PWD/icp-proposal_amd/libicp_proposal_amd_testhooks.so
for B in 10 30; do
  for g in 1 2 3; do
    echo "B=$B groups=$g"; ICP_WIDE_LOOP_GROUPS=$g ICP_HOST_DEVICE_LOOP=1 timeout 600 python3 tools/r5_wide_loop.py face200 $B 60 /tmp/x.npz 2>&1 | tail -1 | cut -c1-120
  done
  echo "B=$B host-stepped"; ICP_HOST_DEVICE_LOOP=0 timeout 600 python3 tools/r5_wide_loop.py face200 $B 60 /tmp/x.npz 2>&1 | tail -1 | cut -c1-120
done
