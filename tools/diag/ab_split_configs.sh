for round in 1 2; do for v in 1 0; do for c in c3b c5; do
ms=$(MI_WGRAD_SPLIT_OLD=$v python3 tools/run_configs.py $c 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
echo "round $round old_split=$v $c $ms ms/step"; done; done; done
