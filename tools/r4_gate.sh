#!/bin/bash
# the gate ahead of the evaluator's search sequence (k_wide_gate): configs[4] rates with and without, then a kernel timeline
for v in 0 1; do
  echo "ICP_WIDE_GATE=$v"; ICP_WIDE_GATE=$v timeout 150 python tools/r4_c4_many.py 3 2>&1 | grep "targets 3"
  ICP_WIDE_GATE=$v timeout 150 python tools/r4_c4_setup.py 10 2>&1 | grep "target [12]"
done
ICP_WIDE_GATE=${TRACE_GATE:-1} timeout 300 bash tools/r4_trace_c4.sh 3
head -${TRACE_LINES:-60} gpurun_out/tr4/timeline.txt
