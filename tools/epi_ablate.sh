#!/bin/bash
# Where the GELU / dGELU epilogues of the fp8 base step spend their time: the product build next to builds without the GELU arithmetic (ATST_EPI_ABL=1),
# without the e4m3 copy of the output (=2), without the u store / load (=4) and without all three (=7); step time of the ATST-base fp8 bench (the numbers
# of the ablated builds are wrong, their timing is not).  Builds (build container):
#   for m in 1 2 4 7; do ATST_LIB_TAG=epiabl$m ATST_EXTRA_FLAGS="-DATST_EPI_ABL=$m" python -c "from audiossl_amd import build; build.build()"; done
for tag in "" epiabl1 epiabl2 epiabl4 epiabl7 ""; do
  echo "== build: ${tag:-product}"
  ATST_LIB_TAG=$tag ${PYTHON:-python} bench.py --arch base --workload clip2 --dtype ${DTYPE:-fp8} --no-cpu-baseline --no-also --steps 20 2>/dev/null | ${PYTHON:-python} -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('  ms/step', d['ms_per_step'], ' '.join(f\"{k['kernel'].replace('gemm_nt_kernel','nt')}={k['avg_us']:.0f}\" for k in d['kernels'] if 'gelu' in k['kernel']))"
done
