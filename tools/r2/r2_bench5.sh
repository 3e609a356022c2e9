mkdir -p gpurun_out/r2i
for st in "20 5" "3000 200"; do set -- $st; timeout 200 python bench.py --steps $1 --warmup $2 --many-chains 0 --cpu-steps 0 --profile-steps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c1', d['steps'], round(d['value']), d['config']['accepted'])"; done
timeout 300 python bench.py --steps 300 --warmup 20 --many-chains 64 --cpu-steps 0 --profile-steps 0 | grep -o "many_chains.*"
timeout 300 python bench.py --steps 300 --warmup 20 --many-chains 32 --cpu-steps 0 --profile-steps 0 | grep -o "many_chains.*"
timeout 400 python bench.py --config 3 --steps 300 --warmup 20 --cpu-steps 0 > gpurun_out/r2i/c3.json 2> gpurun_out/r2i/c3.err; tail -c 300 gpurun_out/r2i/c3.err
timeout 400 python bench.py --config 2 --steps 300 --warmup 20 --cpu-steps 0 > gpurun_out/r2i/c2.json 2> gpurun_out/r2i/c2.err; tail -c 300 gpurun_out/r2i/c2.err
for f in c2 c3; do python - <<PY
import json
d=json.load(open("gpurun_out/r2i/$f.json"))
print("$f", round(d["value"],1), d["config"]["accepted"], d.get("roofline_error"))
print("  kernels:", d.get("kernel_us_per_step"))
PY
done
