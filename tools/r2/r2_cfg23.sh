#!/bin/bash
# configs 2 and 3: face + femur-100 tests, then the benches
mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_gpu_face.py tests/test_gpu_chain.py -m gpu -x -q > gpurun_out/c23/tests.log 2>&1; tail -3 gpurun_out/c23/tests.log
timeout 600 python bench.py --gpus 1 --config 3 --cpu-steps 0 --many-chains 0 > gpurun_out/c23/b3.json 2> gpurun_out/c23/b3.err; tail -c 1500 gpurun_out/c23/b3.json
timeout 600 python bench.py --gpus 1 --config 2 --cpu-steps 0 --many-chains 0 > gpurun_out/c23/b2.json 2> gpurun_out/c23/b2.err; tail -c 600 gpurun_out/c23/b2.json
