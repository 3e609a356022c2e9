#!/bin/bash
# developer A/B on one box: the lockstep runner with and without the helper thread that draws the next step's normals
run() { timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --cpu-steps 0 --profile-steps 0 --many-chains $1 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "iterations/s", "steps_per_chain"' | grep -o '[0-9.]*' | head -1; }
cd icp-proposal_amd/host && g++ -O2 -std=c++17 -fPIC -pthread -ffp-contract=off -fvisibility=hidden -DICP_DEV_SWITCHES -shared -o ../libicp_host.so icp_host.cpp -L.. -licp_proposal_amd -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib && cd ../..
for B in ${CHAINS:-16 32 64 128}; do for k in 1 2 3; do
  echo -n "chains $B, drawn ahead: "; run $B
  echo -n "chains $B, inline:      "; ICP_NO_NORMALS_AHEAD=1 run $B
done; done
make -C icp-proposal_amd/host clean all > /dev/null
