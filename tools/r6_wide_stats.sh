#!/bin/bash
# kernel stats + queue overlap of the 25-chain wide loop at configs[4]'s size (this tree, and _old/ if present)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6wide; mkdir -p $O
export ICP_HOST_DEVICE_LOOP=1
for d in . ${AB:+_old}; do
  tag=$( [ $d = . ] && echo new || echo old )
  (cd $d && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o s -- python3 tools/r5_wide_loop.py facefull ${NCH:-25} 200 /tmp/x.npz > $O/$tag.log 2>&1)
  f=$(find $O/$tag -name '*kernel_stats.csv' | head -1)
  python3 tools/stats_md.py $f "wide loop $tag" > $O/${tag}_kernel_stats.md
  python3 tools/trace_overlap.py $(find $O/$tag -name '*kernel_trace.csv' | head -1) 0.5 > $O/${tag}_queue_overlap.txt
  python3 tools/r6_step_timeline.py $(find $O/$tag -name '*kernel_trace.csv' | head -1) > $O/${tag}_step_timeline.txt
  find $O/$tag -name '*kernel_trace.csv' -delete
  tail -1 $O/$tag.log | cut -c1-100
  head -24 $O/${tag}_kernel_stats.md | cut -c1-160
  head -12 $O/${tag}_queue_overlap.txt | cut -c1-200
done
