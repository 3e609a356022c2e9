set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/nospec -o t -- python3 bench.py --gpus 1 --steps 40 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2b/nospec.json 2> gpurun_out/r2b/nospec.err
ICP_SPECULATION=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/spec -o t -- python3 bench.py --gpus 1 --steps 40 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2b/spec.json 2> gpurun_out/r2b/spec.err
find gpurun_out/r2b -name '*.csv' | xargs ls -la
