#!/bin/bash
# Local-view (M = 26624) GEMM launches: 4-wave blocks two per CU (shipped) | 8-wave 128-row tile, 128 registers, two blocks per CU (ATST_TUNE=360) |
# the same tile compiled for 256 registers, one block per CU (lib tag mi2: ATST_EXTRA_FLAGS=-DATST_MI2_WPS=2, ATST_TUNE=360).  Inside gpurun.
out=${1:-gpurun_out/local_ab}; mkdir -p $out
for rep in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 40 2>/dev/null | grep '^{' > $out/w4_$rep.json
  ATST_TUNE=360 timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 40 2>/dev/null | grep '^{' > $out/mi2r128_$rep.json
  ATST_LIB_TAG=mi2 ATST_TUNE=360 timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 40 2>/dev/null | grep '^{' > $out/mi2r256_$rep.json
done
python - <<PY
import json,glob
for tag in ("w4","mi2r128","mi2r256"):
    v=[json.load(open(f))["value"] for f in sorted(glob.glob("$out/%s_*.json"%tag))]
    print("%-8s %s   mean %.1f clips/s" % (tag, " / ".join("%.1f"%x for x in v), sum(v)/max(len(v),1)))
PY
