#!/bin/bash
# kernel trace of the configs[3] chain: timeline of a few steps in the middle
mkdir -p gpurun_out/tr3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/c3.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
ctx = pkg.IcpContext(model, target, device=0)
setup = pkg.bfm_fitting_partial(model, target, evaluator="hausdorff")
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=5)
ch.run(50, want_records=False)
t0 = time.perf_counter(); rec = ch.run(200); dt = time.perf_counter() - t0
print("rate", 200 / dt, "leaves", rec[100:130, 2].astype(int).tolist(), "acc", rec[100:130, 1].astype(int).tolist())
PY
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tr3 -o c3 --output-format csv -- python3 /tmp/c3.py > gpurun_out/tr3/run.log 2>&1
f=$(find gpurun_out/tr3 -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
python3 tools/timeline.py $f $((n * 6 / 10)) 150 > gpurun_out/tr3/timeline.txt
find gpurun_out/tr3 -name "*kernel_trace.csv" -delete
grep rate gpurun_out/tr3/run.log
cat gpurun_out/tr3/timeline.txt
