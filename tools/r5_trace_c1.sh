#!/bin/bash
# kernel trace of the configs[1] chain's first steps (the driver's window: 5 + 20 steps): timeline of the accepted path
mkdir -p gpurun_out/tr1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/c1.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
pkg = g.load_package()
model, target = pkg.data.synthetic_femur_target()
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
ctx = pkg.IcpContext(model, target, device=0)
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
ch.run(5, want_records=False)
t0 = time.perf_counter(); rec = ch.run(20); dt = time.perf_counter() - t0
print("rate", 20 / dt, "leaves", rec[:, 2].astype(int).tolist(), "acc", rec[:, 1].astype(int).tolist())
PY
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tr1 -o c1 --output-format csv -- python3 /tmp/c1.py > gpurun_out/tr1/run.log 2>&1
f=$(find gpurun_out/tr1 -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
python3 tools/timeline.py $f $((n - 150)) 150 > gpurun_out/tr1/timeline.txt
find gpurun_out/tr1 -name "*kernel_trace.csv" -delete
grep rate gpurun_out/tr1/run.log
cat gpurun_out/tr1/timeline.txt
