#!/bin/bash
# what of a busy chip slows the one-workgroup decomposition kernels down (tools/tri_bench.hip: k_hog)
for h in 1 2 3 4 5; do timeout 120 tools/tri_bench 200 $h 2048 2>&1 | grep "beside\|(cold)"; done
timeout 120 tools/tri_bench 200 2 512 2>&1 | grep "beside"
timeout 120 tools/tri_bench 200 1 512 2>&1 | grep "beside"
