#!/bin/bash
# Where the mel front end spends its time: experiment builds with parts of stft_mel_db_kernel switched off (ATST_MEL_ABL bits, csrc/frontend.hip).
#   build container: bash tools/mel_ablate.sh build      GPU box: bash tools/mel_ablate.sh run
VARS="0 1 2 4 8 16 32 63"
if [ "$1" = build ]; then
  for v in $VARS; do ATST_LIB_TAG=mabl$v ATST_EXTRA_FLAGS="-DATST_MEL_ABL=$v" python -c "from audiossl_amd import build; build.build(verbose=False)"; done
else
  for v in $VARS; do echo "== ATST_MEL_ABL=$v"; ATST_LIB_TAG=mabl$v timeout 120 python tools/mel_probe.py 2>&1 | grep "16 kHz"; done
fi
