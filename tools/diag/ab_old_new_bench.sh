#!/bin/bash
# Same-box A/B of bench.py (C4, 128^3) between this tree and ./_ab_old (see ab_old_new.sh)
for round in 1 2 3; do
  for tree in _ab_old .; do
    ms=$(cd $tree && python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round [$tree] $ms ms/step"
  done
done
