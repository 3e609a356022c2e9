#!/bin/bash
# A/B on one box: this tree against the round-4 tree: the default bench's legs.  Needs the old tree under _old/, built:
#   git worktree add _old aa25e50 && python -c 'import sys; sys.path.insert(0, "_old"); import __graft_entry__ as g; g.build()'   (remove it afterwards: it travels with every gpurun)
cd $GRAFT_REPO_ROOT
leg() { (cd $1 && python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 0 --root-sampler-leg 0 --extra-configs=2,3 --many-chains 64 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['extra_configs']
print('$2 headline %d | config2 %d fit %d | config3 %d | many %d' % (d['value'], e['config2']['value'], e['config2']['from_deterministic_fit']['value'], e['config3']['value'], d['many_chains']['value']))"); }
for rep in 1 2 3; do leg . new; leg _old old; done
