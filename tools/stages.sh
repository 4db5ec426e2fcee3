#!/bin/bash
# 256x384 tile: ring depth sweep (run on the GPU box)
for n in 2 3 4; do
  ATST_TALL_STAGES=$n python audiossl_amd/build.py > /dev/null 2>&1 && echo "TALL_STAGES=$n (tall for all epilogues)" && VARIANT=304 python tools/gemm_bench.py 2>&1 | grep -E " nt "
done
python audiossl_amd/build.py > /dev/null 2>&1
python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -1
