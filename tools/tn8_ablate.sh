#!/bin/bash
# Where a stage of the e4m3 weight-gradient kernel (csrc/gemm_tn8.hip) goes: product build / no fragment reads (ATST_TN8_ABL=1) / no LDS-DMA (=2) / no MFMAs (=4) /
# no atomics (=8), tools/wgrad8_bench.py on each.  Builds (build container):
#   for m in 1 2 4 8; do ATST_LIB_TAG=tn8abl$m ATST_EXTRA_FLAGS="-DATST_TN8_ABL=$m" python -c "from audiossl_amd import build; build.build()"; done
for tag in "" tn8abl1 tn8abl2 tn8abl4 tn8abl8 tn8abl3 ""; do
  echo "== build: ${tag:-product}"
  ATST_LIB_TAG=$tag ${PYTHON:-python} tools/wgrad8_bench.py 2>/dev/null | grep wgrad | sed 's/bf16.*e4m3/e4m3/'
done
