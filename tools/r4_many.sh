#!/bin/bash
for g in 16 32 64; do for t in 1 2 3 4; do echo "ICP_WIDE_GROUP=$g targets=$t"; ICP_WIDE_GROUP=$g python tools/r4_c4_many.py $t 2>&1 | grep "targets"; done; done
