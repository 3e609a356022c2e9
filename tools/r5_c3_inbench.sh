#!/bin/bash
# why is the configs[3] leg inside the default bench slower than `bench.py --config 3`?
cd $GRAFT_REPO_ROOT
c3() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['extra_configs']['config3']['value']))"; }
python3 bench.py --config 3 --steps 600 --warmup 100 --many-chains 0 --cpu-steps 0 --profile-steps 0 --extra-configs= --root-sampler-leg 0 2>/dev/null | python3 -c "import sys,json; print('standalone', round(json.loads(sys.stdin.read())['value']))"
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 0 --root-sampler-leg 0 --extra-configs=3 --many-chains 0 2>/dev/null | c3 "after-headline-only"
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 300 --root-sampler-leg 0 --extra-configs=3 --many-chains 0 2>/dev/null | c3 "after-headline+profile-leg"
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 0 --root-sampler-leg 0 --extra-configs=2,3 --many-chains 0 2>/dev/null | c3 "after-headline+config2"
ICP_NO_POOL=1 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 0 --root-sampler-leg 0 --extra-configs=3 --many-chains 0 2>/dev/null | c3 "after-headline-only, no pools"
