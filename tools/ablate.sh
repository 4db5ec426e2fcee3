#!/bin/bash
# GPU box: rebuild the library with an ablation switch and time GEMM shapes.  usage: tools/ablate.sh "1 3 4 5 6" [variant]
for a in $1; do
  ATST_ABLATE=$a python audiossl_amd/build.py > /dev/null 2>&1 && echo "ABLATE=$a VARIANT=${2:-0}" && VARIANT=${2:-0} EXTRA=8192x1152x6144 timeout 120 python tools/gemm_bench.py 2>&1 | grep -E "qkv fwd|extra"
done
python audiossl_amd/build.py > /dev/null 2>&1
