#!/bin/bash
# dev: kernel statistics of the 64-chain leg for several builds of the library (the shared-target filter: group sizes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ps; mkdir -p $O
for v in "$@"; do
  export ICP_LIBRARY_PATH=$PWD/icp-proposal_amd/$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o s -- python3 bench.py --steps 50 --warmup 5 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0 --many-chains 64 > $O/$v.json 2> $O/$v.err
  f=$(find $O/$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; [ -n "$f" ] && grep -E "filter_batch|regression_batch|resolve_batch|begin_batch" "$f" | cut -d, -f1-4 | cut -c1-150
  find $O/$v -name '*kernel_trace.csv' -delete
done
