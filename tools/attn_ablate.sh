#!/bin/bash
# Where the NP = 256 attention backward spends its time: experiment builds with parts of attn_bwd256_kernel switched off (ATST_ATTN_ABL bits,
# csrc/attention.hip), timed side by side at the bench geometry (512 sequences, 6 heads).
#   build container:  bash tools/attn_ablate.sh build        GPU box:  bash tools/attn_ablate.sh run
VARS="0 1 2 4 8 16 6 22 30"
if [ "$1" = build ]; then
  for v in $VARS; do ATST_LIB_TAG=abl$v ATST_EXTRA_FLAGS="-DATST_ATTN_ABL=$v" python -c "from audiossl_amd import build; build.build(verbose=False)"; done
else
  for v in $VARS; do echo "== ATST_ATTN_ABL=$v"; ATST_LIB_TAG=abl$v timeout 120 python tools/attn_time.py; done
fi
