#!/bin/bash
# Re-measure roofline.traffic for the current kernel sources: four PMC passes + tools/pmc_traffic.py <tag>
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
tag=${1:-r02h}
rm -rf gpurun_out/pmc_fwd_* gpurun_out/pmc_wgrad_*
for k in fwd wgrad; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 -d gpurun_out/pmc_${k}_${c} --output-format csv --kernel-trace --pmc $c -- python3 tools/pmc_conv.py $k 32 32 128 > gpurun_out/pmc_${k}_${c}.log 2>&1; done; done
python3 tools/pmc_traffic.py gpurun_out $tag 128 | tail -6
cp profiles/${tag}_pmc_traffic.json profiles/${tag}_pmc_dispatches.csv gpurun_out/
