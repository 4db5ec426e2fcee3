#!/bin/bash
# Ablation of the 192x384 weight-gradient kernel on the GPU box.  Needs the experiment switches:
#   patch -p0 audiossl_amd/csrc/gemm.hip < tools/experiments/wgrad_schedule_variants.patch      (revert afterwards)
# ATST_TN_ABL: 1 no atomics, 2 no MFMA, 3 no LDS-DMA, 4 no fragment reads, 5 no LDS-DMA + no barrier, 6 no barrier
for a in ${ABLS:-1 2 3 4 5 6}; do
  ATST_EXTRA_FLAGS=-DATST_TN_ABL=$a python audiossl_amd/build.py --force > /dev/null 2>&1
  echo "== ATST_TN_ABL $a"
  ATST_EXTRA_FLAGS=-DATST_TN_ABL=$a CFGS=${CFGS:-0,1,2,3,4,5} timeout 300 python tools/wgrad_cfg.py 2>&1 | grep "M=131072"
done
