set -x
mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/nproc.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --many-chains 0 --cpu-steps 0 > gpurun_out/r2a/b20.json 2> gpurun_out/r2a/b20.err
python bench.py --gpus 1 --steps 20 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2a/b20b.json 2>> gpurun_out/r2a/b20.err
ICP_SPECULATION=1 python bench.py --gpus 1 --steps 20 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2a/b20_spec.json 2>> gpurun_out/r2a/b20.err
ICP_HOST_TIMING=1 python bench.py --gpus 1 --steps 3000 --warmup 200 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2a/b3000.json 2> gpurun_out/r2a/b3000.err
ICP_HOST_TIMING=1 ICP_SPECULATION=1 python bench.py --gpus 1 --steps 3000 --warmup 200 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2a/b3000_spec.json 2> gpurun_out/r2a/b3000_spec.err
ICP_HOST_TIMING=1 python bench.py --gpus 1 --steps 200 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2a/b200.json 2> gpurun_out/r2a/b200.err
ICP_HOST_TIMING=1 ICP_SPECULATION=1 python bench.py --gpus 1 --steps 200 --warmup 5 --many-chains 0 --cpu-steps 0 --profile-steps 0 > gpurun_out/r2a/b200_spec.json 2> gpurun_out/r2a/b200_spec.err
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1
tail -3 gpurun_out/r2a/pytest.log
for f in gpurun_out/r2a/b*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['config']['accepted'], d['config']['icp_proposals'])"; done
