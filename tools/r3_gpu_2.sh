#!/bin/bash
mkdir -p gpurun_out/r3_2
timeout 900 python bench.py --cpu-steps 0 --extra-configs "" > gpurun_out/r3_2/bench.json 2> gpurun_out/r3_2/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_2/bench.json'))
print(d['value'], d.get('cholesky_root_sampler'), d.get('many_chains'))
print(d['kernel_us_per_step']); print(d['roofline']['distance_kernel']); print(d['roofline']['whole_step'].get('executed'))
PY
