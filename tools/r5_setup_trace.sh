cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5setup
python3 tools/r5_setup_cost.py 20 3 > gpurun_out/r5setup/plain.log 2>&1
timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d gpurun_out/r5setup/tr -o s -- python3 tools/r5_setup_cost.py 20 3 > gpurun_out/r5setup/traced.log 2>&1
f=$(find gpurun_out/r5setup/tr -name '*hip_api_stats.csv' | head -1); head -25 $f > gpurun_out/r5setup/hip_api_stats.txt
find gpurun_out/r5setup/tr -name '*trace.csv' -delete
cat gpurun_out/r5setup/plain.log; cat gpurun_out/r5setup/hip_api_stats.txt
