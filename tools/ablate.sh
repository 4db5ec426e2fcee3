#!/bin/bash
# GPU box: rebuild the library with experiment switches and time the GEMM shapes.
# usage: tools/ablate.sh "BK:NSTAGE:ABLATE ..."   e.g. "32:2:0 64:2:0 64:2:3"
for cfg in $1; do
  IFS=: read bk ns ab <<< "$cfg"
  ATST_BK=$bk ATST_NSTAGE=$ns ATST_ABLATE=$ab python audiossl_amd/build.py > /dev/null 2>&1 && echo "BK=$bk NSTAGE=$ns ABLATE=$ab" && timeout 120 python tools/gemm_bench.py 2>&1 | grep -E " nt "
done
python audiossl_amd/build.py > /dev/null 2>&1
