# A/B of the lockstep runner's schedule on one box: groups in flight x fifth launch on its own stream or not
for rep in 1 2; do
for g in 1 2 3; do for f in 0 1; do
  if [ $f = 1 ]; then export ICP_BATCH_FINISH_INLINE=1; else unset ICP_BATCH_FINISH_INLINE; fi
  echo "== groups $g finish_inline $f: $(ICP_LOCKSTEP_GROUPS=$g timeout 500 python tools/multichain.py batched ${1:-16,32} 2>&1 | grep 'B=' | awk '{printf "B=%s %s it/s  ", $3, $4}')"
done; done; done
