#!/bin/bash
# the tridiagonalisation's load prologue: stand-alone timing, parity tests, configs[4] rates
for r in 100 150 200; do timeout 120 tools/tri_bench $r 2>&1 | grep "us per decomposition (cold)\|status"; done
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_face.py -x -q -m gpu 2>&1 | tail -3
bash tools/r4_gate.sh
timeout 300 python bench.py --config 3 --steps 600 --warmup 50 --many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0 2>&1 | grep -o '"value": [0-9.]*' | head -1
