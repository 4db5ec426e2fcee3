#!/bin/bash
# tall wgrad: issue placement x atomics (GPU box)
for i in 1 2; do for a in 8; do
  ATST_TN_ISSUE=$i ATST_ABLATE=$a python audiossl_amd/build.py > /dev/null 2>&1 && echo "ISSUE=$i ABLATE=$a" && python tools/gemm_bench.py 2>&1 | grep -E " tn "
done; done
