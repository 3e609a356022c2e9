#!/bin/bash
# A/B of developer switches on the driver's window (5 + 20 steps of configs[1]), alternating runs on one box
cd $GRAFT_REPO_ROOT
B="--many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0"
export ICP_LIBRARY_PATH=$PWD/icp-proposal_amd/libicp_proposal_amd_testhooks.so
for rep in 1 2 3 4; do
  for v in "base" "$@"; do
    if [ "$v" = "base" ]; then e=""; else e="$v"; fi
    r=$(env $e python3 bench.py --steps 20 --warmup 5 $B | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
    echo "$v: $r"
  done
done
