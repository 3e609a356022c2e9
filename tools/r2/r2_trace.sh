cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2j/t20 -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2j/t20.json 2> gpurun_out/r2j/t20.err
