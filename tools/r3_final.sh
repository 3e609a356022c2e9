mkdir -p gpurun_out/final3
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/final3/pytest.log 2>&1; tail -3 gpurun_out/final3/pytest.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/final3/smoke.log 2>&1; tail -4 gpurun_out/final3/smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final3/bench.json 2> gpurun_out/final3/bench.err; tail -3 gpurun_out/final3/bench.err
python3 bench.py --config 3 > gpurun_out/final3/bench_c3.json 2> gpurun_out/final3/bench_c3.err
python3 bench.py --config 4 --steps 50 > gpurun_out/final3/bench_c4.json 2> gpurun_out/final3/bench_c4.err
python3 - <<'PY'
import json
for f in ("bench","bench_c3","bench_c4"):
    try:
        d=json.loads(open("gpurun_out/final3/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), d["unit"], {k:round(v["value"],1) for k,v in d.get("extra_configs",{}).items() if "value" in v}, round(d.get("many_chains",{}).get("value",0)), d.get("runtime_stats"))
    except Exception as e: print(f,"ERR",e)
PY
