#!/bin/bash
# Same-box A/B of bench.py under environment knobs: ab_bench.sh "VAR=val VAR2=val" "VAR=val" ...   (interleaved, 2 rounds)
for round in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % j['ms_per_step'])")
    echo "round $round [$cfg] $ms ms/step"
  done
done
