#!/bin/bash
# two host threads, each with a batch of chains (tools/multichain.py batched2), each size in a process of its own, several times
for rep in 1 2 3; do
  for b in 4 16 32; do
    echo "rep $rep B=$b: $(timeout 90 python tools/multichain.py batched2 $b 2>&1 | tail -2 | tr '\n' ' ')"
  done
done
