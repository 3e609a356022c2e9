cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in ${MACROS:-1 2}; do
  export ICP_LIBRARY_PATH=$PWD/icp-proposal_amd/libicp_proposal_amd_testhooks.so ICP_REGRESSION_MACRO=$m
  rm -rf gpurun_out/rs$m; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rs$m -o s -- python3 tools/r4_c4_many.py 3 > /dev/null 2>&1
  f=$(find gpurun_out/rs$m -name '*kernel_stats.csv' | head -1)
  echo "macro $m: $(grep -E 'k_wide_regression|k_sum_partials' $f | cut -d, -f1-4 | tr '\n' ' ')"
  find gpurun_out/rs$m -name '*trace.csv' -delete
done
