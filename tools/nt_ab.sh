#!/bin/bash
# Cache policy of the GEMM epilogue traffic, same box, alternating (inside gpurun).  Builds (build container):
#   for m in 14 0 7 11; do ATST_LIB_TAG=nt$m ATST_EXTRA_FLAGS="-DATST_NT=$m" python -c "from audiossl_amd import build; build.build()"; done
# mask bits (csrc/gemm.hip ATST_NT): 0 bf16 stores, 1 fp32 stores of epilogue8, 2 fp32 row stores of the row-wise epilogues, 3 epilogue loads; 15 = product
out=${1:-gpurun_out/nt_ab}; mkdir -p $out
for rep in 1 2 3; do
  for tag in ${TAGS:-prod nt14 nt0 nt7 nt11}; do
    t=$tag; [ $tag = prod ] && t=""
    ATST_LIB_TAG=$t timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 40 2>/dev/null | grep '^{' > $out/${tag}_$rep.json
  done
done
python - <<PY
import json,glob
for tag in "${TAGS:-prod nt14 nt0 nt7 nt11}".split():
    v=[json.load(open(f))["value"] for f in sorted(glob.glob("$out/%s_*.json"%tag))]
    print("%-6s %s   mean %.1f clips/s" % (tag, " / ".join("%.1f"%x for x in v), sum(v)/max(len(v),1)))
PY
