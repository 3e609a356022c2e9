mkdir -p gpurun_out/r2g
timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 --many-chains 0 > gpurun_out/r2g/c1_20.json 2> gpurun_out/r2g/c1_20.err; tail -c 600 gpurun_out/r2g/c1_20.err
timeout 300 python bench.py --config 2 --steps 200 --warmup 20 --cpu-steps 4 > gpurun_out/r2g/c2.json 2> gpurun_out/r2g/c2.err; tail -c 600 gpurun_out/r2g/c2.err
timeout 400 python bench.py --config 3 --steps 200 --warmup 20 --cpu-steps 2 > gpurun_out/r2g/c3.json 2> gpurun_out/r2g/c3.err; tail -c 600 gpurun_out/r2g/c3.err
for f in c1_20 c2 c3; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2g/$f.json"))
except Exception as e:
    print("$f", "no line", e); raise SystemExit
print("$f", round(d["value"],1), d["config"]["accepted"], d.get("roofline_error"), d.get("cpu_baseline_error"))
r=d.get("roofline") or {}
print("  roofline:", r.get("kernel"), r.get("avg_launch_us"), r.get("achieved"), r.get("frac"), (r.get("whole_step") or {}).get("hbm_frac"), (r.get("whole_step") or {}).get("flops_frac_f32_vector_peak"), (r.get("distance_kernel") or {}))
c=d.get("cpu_baseline") or {}
print("  cpu:", c.get("value"), (c.get("B1") or {}).get("sample"), (c.get("B2") or {}).get("value"), (c.get("B2") or {}).get("cores"), (c.get("B2") or {}).get("sample"), c.get("gpu_matches_oracle_on_sample"))
print("  kernels:", d.get("kernel_us_per_step"))
PY
done
