#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in none femur_closed; do
  rm -rf gpurun_out/rot_$m; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rot_$m -o s -- python3 tools/r5_c3_rotate2.py $m > gpurun_out/rot_$m.log 2>&1
  f=$(find gpurun_out/rot_$m -name '*kernel_stats.csv' | head -1)
  echo "== $m: $(tail -1 gpurun_out/rot_$m.log)"; head -14 $f | cut -d, -f1-4 | sed 's/icp::(anonymous namespace):://; s/icp::tri:://' | cut -c1-110
  find gpurun_out/rot_$m -name '*trace.csv' -delete
done
