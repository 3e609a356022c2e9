cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2f/t20 -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2f/t20.json 2> gpurun_out/r2f/t20.err
ICP_HOST_TIMING=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 2>&1 | grep "icp "
