#!/bin/bash
# Where do the phase kernels (convph.hip) spend their time?  Same launches with parts compiled / switched out (results are garbage):
#   base | no store epilogue (MI_CPH_DBG=1: read by the diagnostic builds only, PHBASE = nothing compiled out) | no weight LDS-DMA (PHW) | no halo LDS-DMA (PHHALO) | neither (PHDMA) | no s_barrier (PHNOBAR)
# needs: make -C medical_image_generation_amd/csrc phdiag
D=medical_image_generation_amd/diag
echo "== base"; python3 tools/bench_ph.py
echo "== no store epilogue"; MI_CPH_DBG=1 MI_LIB_PATH=$PWD/$D/libmedimgen_hip_PHBASE.so python3 tools/bench_ph.py
for v in W HALO DMA NOBAR; do
  echo "== $v"; MI_LIB_PATH=$PWD/$D/libmedimgen_hip_PH$v.so python3 tools/bench_ph.py
  echo "== $v + no store epilogue"; MI_CPH_DBG=1 MI_LIB_PATH=$PWD/$D/libmedimgen_hip_PH$v.so python3 tools/bench_ph.py
done
