#!/bin/bash
# developer A/B: lockstep groups x chains per GPU (builds the host harness with its developer switches, restores it afterwards)
cd icp-proposal_amd/host && g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -pthread -DICP_DEV_SWITCHES -shared -o ../libicp_host.so icp_host.cpp -L.. -licp_proposal_amd -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib && cd ../..
for B in ${CHAINS:-32 48 96 128}; do
for g in ${GROUPS_:-2 3 4}; do
  for k in 1 2; do
    echo -n "chains $B groups $g: "; ICP_LOCKSTEP_GROUPS=$g timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --cpu-steps 0 --profile-steps 0 --many-chains $B 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "iterations/s", "steps_per_chain"' | grep -o '[0-9.]*' | head -1
  done
done
done
make -C icp-proposal_amd/host clean all > /dev/null
