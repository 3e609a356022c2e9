#!/bin/bash
# the on-device wide loop against host-stepped chains over model kinds and chain counts (records compared by tools/r5_wide_loop.py)
cd $GRAFT_REPO_ROOT
for spec in "face100 3 40" "face100 7 40" "face200 4 30" "hausdorff 5 40" "facefull 2 40" "facefull 9 60" "femur100 3 40" "femur200 2 30" "face40 6 60" "facefull 17 40"; do
  set -- $spec
  echo "== $spec: $(timeout 900 python3 tools/r5_wide_loop.py $1 $2 $3 2>&1 | grep -E 'identical|DIFFERENT|failed' | tr '\n' ' ')"
done
