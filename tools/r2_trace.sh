cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2e/t40 -o t -- python3 bench.py --gpus 1 --steps 40 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2e/t40.json 2> gpurun_out/r2e/t40.err
ICP_HOST_TIMING=1 python3 bench.py --gpus 1 --steps 40 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 2>&1 | grep "icp host"
ICP_HOST_TIMING=1 python3 bench.py --gpus 1 --steps 3000 --warmup 200 --many-chains 0 --cpu-steps 0 --profile-steps 0 2>&1 | grep "icp host"
