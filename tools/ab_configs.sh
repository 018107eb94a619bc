#!/bin/bash
# Same-box A/B of tools/run_configs.py under environment knobs: ab_configs.sh CONFIG "VAR=val" "VAR=val VAR2=val" ...  (interleaved, 2 rounds)
cfgname=$1; shift
for round in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg python3 tools/run_configs.py $cfgname 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round $cfgname [$cfg] $ms ms/step"
  done
done
