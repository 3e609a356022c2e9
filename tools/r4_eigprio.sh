#!/bin/bash
# configs[4], 30 chains of 3 targets per submission: the decompositions on the launch context's own (greatest-priority) pair of streams
# against the default-priority pool; and 10 chains of one target
for v in 0 1; do
  echo "ICP_BATCH_EIG_OWN=$v"; ICP_BATCH_EIG_OWN=$v python tools/r4_c4_many.py 3 2>&1 | grep "targets 3"
  ICP_BATCH_EIG_OWN=$v python tools/r4_c4_setup.py 10 2>&1 | grep "target [12]"
done
echo "ICP_STREAM_PRIORITY=0"; ICP_STREAM_PRIORITY=0 python tools/r4_c4_many.py 3 2>&1 | grep "targets 3"
