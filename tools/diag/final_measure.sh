set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > gpurun_out/r2f_tests.log 2>&1; tail -3 gpurun_out/r2f_tests.log
python3 bench.py > gpurun_out/r02h_bench128.json 2> gpurun_out/r02h_bench128.err; tail -c 1500 gpurun_out/r02h_bench128.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02h -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02h_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02h_roof -- python3 bench.py --roofline-only > gpurun_out/r02h_roofline_only.json 2> gpurun_out/r02h_roof.err
for k in fwd wgrad; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 -d gpurun_out/pmc_${k}_${c} --output-format csv --kernel-trace --pmc $c -- python3 tools/pmc_conv.py $k 32 32 128 > gpurun_out/pmc_${k}_${c}.log 2>&1; done; done
python3 tools/pmc_traffic.py gpurun_out r02h 128 > gpurun_out/r02h_pmc.log 2>&1; tail -5 gpurun_out/r02h_pmc.log
cp profiles/r02h_pmc_traffic.json profiles/r02h_pmc_dispatches.csv gpurun_out/
for c in c2 c3a c3b c3b_ldm c5 c5_ckpt; do python3 tools/run_configs.py $c 10 2>/dev/null | tail -1; done > gpurun_out/r02h_configs.json; cat gpurun_out/r02h_configs.json | cut -c1-120
python3 tools/bench_sample.py 128 50 2>/dev/null | tail -1 > gpurun_out/r02h_sampling.json; cat gpurun_out/r02h_sampling.json | cut -c1-200
