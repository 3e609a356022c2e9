#!/bin/bash
# kernel stats of the wide step's on-device loop (30 chains of the face configuration, one group / three groups)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/wlprof; mkdir -p $O
export ICP_LIBRARY_PATH=$PWD/icp-proposal_amd/libicp_proposal_amd_testhooks.so
export ICP_HOST_DEVICE_LOOP=1
for g in ${1:-1}; do
  export ICP_WIDE_LOOP_GROUPS=$g
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$g -o s -- python3 tools/r5_wide_loop.py ${KIND:-face200} ${NCH:-30} ${NST:-60} /tmp/x.npz > $O/g$g.log 2>&1
  f=$(find $O/g$g -name '*kernel_stats.csv' | head -1)
  python3 tools/stats_md.py $f "wide loop, 30 chains, $g group(s)" > $O/wl_g${g}_kernel_stats.md
  python3 tools/trace_overlap.py $(find $O/g$g -name '*kernel_trace.csv' | head -1) 0.5
  find $O/g$g -name '*kernel_trace.csv' -delete
  tail -2 $O/g$g.log | cut -c1-100
  head -16 $O/wl_g${g}_kernel_stats.md | cut -c1-150
done
