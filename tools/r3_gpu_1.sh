#!/bin/bash
# round 3, GPU call 1: full GPU suite, time-out repro (three arms), default bench, configs 2/3/4 benches
mkdir -p gpurun_out/r3_1
python -m pytest tests -m gpu -x -q > gpurun_out/r3_1/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_1/pytest.log; tail -5 gpurun_out/r3_1/pytest.log
H=icp-proposal_amd/libicp_proposal_amd_testhooks.so
for rep in 1 2 3; do
ARM=r2   ICP_LIBRARY_PATH=$H ICP_TEST_EIGEN_CHUNK=24 ICP_TEST_NO_GATE=1 timeout 300 python tools/r3_timeout_repro.py 2>&1 | tail -1
ARM=gate ICP_LIBRARY_PATH=$H ICP_TEST_EIGEN_CHUNK=24 timeout 300 python tools/r3_timeout_repro.py 2>&1 | tail -1
ARM=r3   timeout 300 python tools/r3_timeout_repro.py 2>&1 | tail -1
done | tee gpurun_out/r3_1/repro.log
timeout 900 python bench.py > gpurun_out/r3_1/bench_default.json 2> gpurun_out/r3_1/bench_default.err; echo "bench rc=$?"; head -c 1500 gpurun_out/r3_1/bench_default.json; echo
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_1/bench_20.json 2> gpurun_out/r3_1/bench_20.err; python -c "
import json; d=json.load(open('gpurun_out/r3_1/bench_20.json')); print('20-step', d['value'], d.get('extra_configs'), d.get('many_chains'), d.get('runtime_stats'))"
timeout 900 python bench.py --config 4 --steps 50 --warmup 5 > gpurun_out/r3_1/bench_c4.json 2> gpurun_out/r3_1/bench_c4.err; echo "c4 rc=$?"; head -c 1200 gpurun_out/r3_1/bench_c4.json; echo
