#!/bin/bash
# tri_bench on the two routes, several ranks
mkdir -p gpurun_out
for r in ${RANKS:-200 101 51 140 64 65 128 129 192 193 7 3}; do
  for t in ${ROUTES:-1 0}; do
    echo "== rank $r ICP_EIGEN_TRIDIAG=$t"
    ICP_EIGEN_TRIDIAG=$t timeout 120 tools/tri_bench $r
  done
done
