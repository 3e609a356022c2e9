#!/bin/bash
# A/B (this tree / _old): wide loop 25, the 64-chain merged loop, 10-chain face loop, configs[4] 10x10x300
cd $GRAFT_REPO_ROOT
leg() { (cd $1
   w=$(ICP_HOST_DEVICE_LOOP=1 timeout 600 python3 tools/r5_wide_loop.py facefull 25 200 /tmp/x_$2.npz 2>&1 | tail -1 | awk '{print $4, $5}')
   w10=$(ICP_HOST_DEVICE_LOOP=1 timeout 600 python3 tools/r5_wide_loop.py facefull 10 200 /tmp/y_$2.npz 2>&1 | tail -1 | awk '{print $4, $5}')
   m=$(ICP_HOST_DEVICE_LOOP=1 timeout 600 python3 tools/r3_device_loop.py 64 600 eigen /tmp/z_$2.npy 2>&1 | tail -1 | grep -o "rate=[0-9.]* it/s\|[0-9.]* it/s" | head -1)
   c4=$(python3 bench.py --config 4 --steps 300 --warmup 5 2>/dev/null | python3 -c "import sys,json; print('%d' % json.loads(sys.stdin.read())['value'])")
   echo "$2: wide25 $w | wide10 $w10 | merged64 $m | config4 10x10x300 $c4")
}
for rep in $(seq 1 ${1:-1}); do leg . new; leg _old old; done
python3 - <<'PY'
import numpy as np
for f in ('x', 'y'):
    a, b = np.load('/tmp/%s_new.npz' % f), np.load('/tmp/%s_old.npz' % f)
    print(f, 'wide loop records identical:', all(np.array_equal(a[k], b[k]) for k in ('a', 'single', 'b')))
PY
