#!/bin/bash
# Same-box A/B of a config between this tree and a worktree of an older commit at ./_ab_old (git worktree add _ab_old <commit>; make).
# usage: bash tools/diag/ab_old_new.sh c3b 10
cfg=${1:-c3b}; steps=${2:-10}
for round in 1 2; do
  for tree in _ab_old .; do
    ms=$(cd $tree && python3 tools/run_configs.py $cfg $steps 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round [$tree] $cfg $ms ms/step"
  done
done
