#!/bin/bash
# MFMA-pipe and LDS activity of the two dominant conv kernels from hardware counters (separate passes, program directly after `--`)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for shape in "32 32" "64 64"; do
  tag=$(echo $shape | tr ' ' '_')
  for k in fwd wgrad; do
    rocprofv3 -d gpurun_out/pmcu_${k}_${tag}_mfma --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/pmc_conv.py $k $shape 128 > gpurun_out/pmcu_${k}_${tag}_mfma.log 2>&1
    rocprofv3 -d gpurun_out/pmcu_${k}_${tag}_lds --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -- python3 tools/pmc_conv.py $k $shape 128 > gpurun_out/pmcu_${k}_${tag}_lds.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, re
out = []
for d in sorted(glob.glob('gpurun_out/pmcu_*')):
    if not d.endswith(('_mfma', '_lds')): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'k_conv27<[^>]*>|k_conv_wgrad2<[^>]*>', r['Kernel_Name'])
            if m: acc[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        line = f"{d.split('/')[-1]:28s} {k:24s} " + "  ".join(f"{c}={sum(v[1:])/max(len(v)-1,1):.4g}" for c, v in sorted(cs.items()))
        out.append(line)
print("\n".join(out))
PY
