cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cs -o s -- python3 tools/r5_setup_cost.py 12 3 > /dev/null 2>&1
f=$(find gpurun_out/cs -name '*kernel_stats.csv' | head -1); head -16 $f | cut -d, -f1-5 | cut -c1-150
find gpurun_out/cs -name '*trace.csv' -delete
