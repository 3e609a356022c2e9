mkdir -p gpurun_out/r2h
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --config 2 --steps 300 --warmup 20 --cpu-steps 0 > gpurun_out/r2h/c2.json 2> gpurun_out/r2h/c2.err; tail -c 400 gpurun_out/r2h/c2.err
timeout 400 python bench.py --config 3 --steps 300 --warmup 20 --cpu-steps 0 > gpurun_out/r2h/c3.json 2> gpurun_out/r2h/c3.err; tail -c 400 gpurun_out/r2h/c3.err
for f in c2 c3; do python - <<PY
import json
d=json.load(open("gpurun_out/r2h/$f.json"))
print("$f", round(d["value"],1), d["config"]["accepted"], d.get("roofline_error"))
r=d.get("roofline") or {}
print("  roofline:", r.get("kernel"), r.get("avg_launch_us"), r.get("frac"))
print("  kernels:", d.get("kernel_us_per_step"))
PY
done
