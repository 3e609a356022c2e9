#!/bin/bash
# full GPU test suite, then the three configurations' bench lines
mkdir -p gpurun_out/full
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/full/tests.log 2>&1; tail -4 gpurun_out/full/tests.log
timeout 600 python bench.py --gpus 1 --config 3 --cpu-steps 0 --many-chains 0 > gpurun_out/full/b3.json 2> gpurun_out/full/b3.err; grep -o '"value": [0-9.]*' gpurun_out/full/b3.json | head -1
timeout 600 python bench.py --gpus 1 --config 2 --cpu-steps 0 --many-chains 0 > gpurun_out/full/b2.json 2> gpurun_out/full/b2.err; grep -o '"value": [0-9.]*' gpurun_out/full/b2.json | head -1
timeout 600 python bench.py --gpus 1 --cpu-steps 0 --many-chains 0 > gpurun_out/full/b1.json 2> gpurun_out/full/b1.err; grep -o '"value": [0-9.]*' gpurun_out/full/b1.json | head -1
