#!/bin/bash
# k_conv27 ablation: which part of the kernel costs the cycles between the 32 an MFMA needs and the 54-70 it gets?
# Needs `make -C medical_image_generation_amd/csrc diag`.  usage: bash tools/diag/c27_ablate.sh > gpurun_out/c27_ablate.log 2>&1
cd "$(dirname "$0")/../.."
D=medical_image_generation_amd/diag
for v in ${VARIANTS:-base A B AB HALO W DMA NOBAR ALL HOT PF1 PF2 PF3}; do
  for dbg in ${DBGS:-0 1}; do
    if [ $v = base ]; then export MI_LIB_PATH=$PWD/$D/libmedimgen_hip_BASE.so; else export MI_LIB_PATH=$PWD/$D/libmedimgen_hip_$v.so; fi  # (the MI_C27_DBG knob only exists in the diagnostic builds)
    MI_C27_DBG=$dbg python3 tools/diag/c27_ablate.py 10 || exit 1
    MI_C27_DBG=$((dbg + 64)) python3 tools/diag/c27_ablate.py 1 2>&1 | grep "conv27<" | awk '{k=$1 $2; last[k]=$0} END {for (k in last) print last[k]}' | sort || true
  done
done
