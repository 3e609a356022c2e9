#!/bin/bash
run() { timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --cpu-steps 0 --profile-steps 0 --many-chains $1 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "iterations/s", "steps_per_chain"' | grep -o '[0-9.]*' | head -1; }
for B in 32 64; do for k in 1 2; do echo -n "product lib, chains $B: "; run $B; done; done
cd icp-proposal_amd/host && g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DICP_DEV_SWITCHES -shared -o ../libicp_host.so icp_host.cpp -L.. -licp_proposal_amd -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib && cd ../..
for B in 32 64; do for g in 0 2 3; do echo -n "dev lib, chains $B groups $g: "; ICP_LOCKSTEP_GROUPS=$g run $B; done; done
make -C icp-proposal_amd/host clean all > /dev/null
