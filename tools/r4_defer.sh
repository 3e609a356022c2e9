#!/bin/bash
# configs[4] steady state: waiting chains sit out a round (default) against strict lockstep, one and two groups
for env in "" "ICP_NO_DEFERRAL=1" "ICP_LOCKSTEP_GROUPS=2" "ICP_LOCKSTEP_GROUPS=2 ICP_NO_DEFERRAL=1"; do
  echo "env: $env"; env $env python tools/r4_c4_setup.py 10 2>&1 | grep "target [12]"
done
