#!/bin/bash
mkdir -p gpurun_out/r3_4
python -m pytest tests -m gpu -x -q > gpurun_out/r3_4/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_4/pytest.log
timeout 900 python bench.py > gpurun_out/r3_4/bench_default.json 2> gpurun_out/r3_4/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_4/bench_20.json 2> gpurun_out/r3_4/bench_20.err
python - <<'PY'
import json
for f in ("bench_default", "bench_20"):
    d = json.load(open("gpurun_out/r3_4/%s.json" % f))
    print(f, round(d["value"]), "many", d.get("many_chains", {}).get("value"), "root", {k: v for k, v in d.get("cholesky_root_sampler", {}).items() if k in ("value", "first_20_steps")},
          "extra", {k: round(v.get("value", 0)) for k, v in d.get("extra_configs", {}).items()}, d.get("runtime_stats"))
PY
bash tools/r3_profiles.sh 2>&1 | tail -15
