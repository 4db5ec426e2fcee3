#!/bin/bash
# Where a tile's time goes in the 256 x 256 phased GEMM (csrc/gemm_p8.h): product build / no epilogue at all (ATST_P8_ABL=1) / staging + read-back
# without global traffic (=2), base shapes at M = 131072.  Builds (build container):
#   for m in 1 2; do ATST_LIB_TAG=p8abl$m ATST_EXTRA_FLAGS="-DATST_P8_ABL=$m" python -c "from audiossl_amd import build; build.build()"; done
for tag in "" p8abl1 p8abl2; do
  echo "== build: ${tag:-product}"
  ATST_LIB_TAG=$tag VARIANTS="391" BLAS=0 ${PYTHON:-python} tools/gemm_bench_base.py 2>/dev/null | grep -v "^M="
done
