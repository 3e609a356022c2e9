# A/B of the surface filter's query chunk (workgroups per sphere block) on one box: single chain and 32 chains
for rep in 1 2 3; do
for c in 64 128 256 512; do
  s=$(ICP_SURFACE_CHUNK=$c python bench.py --cpu-steps 0 --profile-steps 0 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
  b=$(ICP_SURFACE_CHUNK=$c python tools/multichain.py batched 32 2>&1 | grep 'B=' | awk '{print $3}')
  echo "chunk $c: single $s it/s, 32 chains $b it/s"
done; done
