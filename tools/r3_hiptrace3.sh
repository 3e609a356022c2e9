#!/bin/bash
# host-side API calls of the configs[3] chain beside its kernels: rocprofv3 --kernel-trace --hip-runtime-trace, merged timeline of
# two steps in the middle of the run (tools/timeline2.py).  /tmp/c3.py comes from tools/r3_trace3.sh's here-document.
mkdir -p gpurun_out/ht3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
sed -n '/^cat > \/tmp\/c3.py/,/^PY$/p' tools/r3_trace3.sh | sed '1d;$d' > /tmp/c3.py
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --stats -d gpurun_out/ht3 -o c3 --output-format csv -- python3 /tmp/c3.py ${1:-cholesky-root} > gpurun_out/ht3/run.log 2>&1
grep rate gpurun_out/ht3/run.log
kf=$(find gpurun_out/ht3 -name "*kernel_trace.csv" | head -1)
af=$(find gpurun_out/ht3 -name "*hip_api_trace.csv" | head -1)
n=$(wc -l < $kf)
python3 tools/timeline2.py $kf $af $((n * 6 / 10)) 2500 > gpurun_out/ht3/timeline2_${1:-cholesky-root}.txt
find gpurun_out/ht3 -name "*trace.csv" -delete
wc -l gpurun_out/ht3/timeline2_${1:-cholesky-root}.txt
