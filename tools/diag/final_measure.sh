# One pass over the final sources: tests, bench line, rocprofv3 kernel stats (two step counts: which nodes are capture-time only),
# roofline-only run, the four PMC passes behind roofline.traffic, the other BASELINE configs, sampling.  usage: bash tools/diag/final_measure.sh TAG
set -o pipefail
TAG=${1:-r03v}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${TAG}_bench128.json 2> gpurun_out/${TAG}_bench128.err; tail -c 1200 gpurun_out/${TAG}_bench128.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_s15 -- python3 bench.py --steps 15 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof_s15.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_roof -- python3 bench.py --roofline-only > gpurun_out/${TAG}_roofline_only.json 2> gpurun_out/${TAG}_roof.err
for k in fwd wgrad; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 -d gpurun_out/pmc_${k}_${c} --output-format csv --kernel-trace --pmc $c -- python3 tools/pmc_conv.py $k 32 32 128 > gpurun_out/pmc_${k}_${c}.log 2>&1; done; done
python3 tools/pmc_traffic.py gpurun_out ${TAG} 128 > gpurun_out/${TAG}_pmc.log 2>&1; tail -5 gpurun_out/${TAG}_pmc.log
cp profiles/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_dispatches.csv gpurun_out/
for f in gpurun_out/prof_${TAG}/*/*kernel_stats.csv; do cp $f gpurun_out/${TAG}_bench128_graph_kernel_stats.csv; done
for f in gpurun_out/prof_${TAG}_s15/*/*kernel_stats.csv; do cp $f gpurun_out/${TAG}_bench128_steps15_kernel_stats.csv; done
for f in gpurun_out/prof_${TAG}_roof/*/*kernel_stats.csv; do cp $f gpurun_out/${TAG}_roofline_only_kernel_stats.csv; done
for c in c2 c3a c3a_gan c3b c3b_ldm c5 c5_ckpt; do python3 tools/run_configs.py $c 10 2>/dev/null | tail -1; done > gpurun_out/${TAG}_configs.json; cut -c1-110 gpurun_out/${TAG}_configs.json
python3 tools/bench_sample.py 128 50 2>/dev/null | tail -1 > gpurun_out/${TAG}_sampling.json; cut -c1-200 gpurun_out/${TAG}_sampling.json
# the bench line again with the traffic record of THIS binary in place
python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench128_with_traffic.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('gpurun_out/${TAG}_bench128_with_traffic.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline_fwd']['frac'],d['roofline_fwd']['traffic'])"
