python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for mode in "X=1" "ICP_SPECULATION=0" "ICP_SPECULATION=1"; do
  echo "== mode [$mode]"
  for st in "20 5" "200 5" "3000 200"; do
    set -- $st
    env $mode python bench.py --gpus 1 --steps $1 --warmup $2 --many-chains 0 --cpu-steps 0 --profile-steps 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['steps'], round(d['value']), d['config']['accepted'])"
  done
done
