#!/bin/bash
# row-384 tile ablation (run on the GPU box): 0 full / 1 no stores / 3 loads+ds_read / 4 ds_read+MFMA / 5 loads only / 6 MFMA only / 7 epilogue only
for a in ${ABL:-0 1 3 4 5 6 7}; do
  ATST_ABLATE=$a python audiossl_amd/build.py > /dev/null 2>&1 && echo "ABLATE=$a" && VARIANT=${V:-304} python tools/gemm_bench.py 2>&1 | grep -E " nt (qkv fwd|fc1 dgrad)"
done
ATST_ABLATE=0 python audiossl_amd/build.py > /dev/null 2>&1
