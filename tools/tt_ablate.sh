#!/bin/bash
# Experiment builds of the two-team persistent GEMM (csrc/gemm_tt.h) next to the product library, then tools/tt_bench.py on each.
# Build here (CPU container; the tagged .so files travel with the snapshot):  tools/tt_ablate.sh build     Run on the GPU box:  tools/tt_ablate.sh run
# VARIANTS: "tag:flags" pairs; ATST_TT_ABL bits 1 = EP role idle, 4 = no fragment reads / MFMAs ; ATST_TT_PRIO 0 / 1 / 2 = s_setprio none / ML / EP ; ATST_TT_NO_TOUCH = no L2 prefetch ; ATST_NT = store cache-policy mask
# (v1 of the kernel also had ATST_TT_ISS = who issues the stream and ATST_TT_ABL bit 8 = whole-line sources: git history, profiles/r06_tt_v1_*)
set -e
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-abl1:-DATST_TT_ABL=1 abl4:-DATST_TT_ABL=4 abl5:-DATST_TT_ABL=5 prio0:-DATST_TT_PRIO=0 prio2:-DATST_TT_PRIO=2 notouch:-DATST_TT_NO_TOUCH nt0:-DATST_NT=0}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do t=${v%%:*}; f=${v#*:}; ATST_LIB_TAG=tt$t ATST_EXTRA_FLAGS="${f//,/ }" python3 audiossl_amd/build.py > /dev/null; echo built $t "${f//,/ }"; done
else
  echo "== product"; HOOKS=2000,2001 python3 tools/tt_bench.py ${WHICH:-small} 2>&1 | grep -v amdgpu.ids
  for v in $VARIANTS; do t=${v%%:*}; echo "== $v"; ATST_LIB_TAG=tt$t HOOKS=2001 python3 tools/tt_bench.py ${WHICH:-small} 2>&1 | grep -v amdgpu.ids || true; done
fi
