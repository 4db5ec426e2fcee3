#!/bin/bash
# Same box, same gpurun call, alternating bench.py runs under different environments (tuning hooks, library tags).
# usage (inside gpurun): CONFIGS="base: tt:ATST_TUNE=2001" WORKLOADS="clip6 clip2" [ARGS="--arch base"] bash tools/env_ab.sh <outdir>
out=${1:-gpurun_out/env_ab}; mkdir -p $out
CONFIGS=${CONFIGS:-"base: tt:ATST_TUNE=2001"}; WORKLOADS=${WORKLOADS:-clip6}; REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
  for w in $WORKLOADS; do
    for c in $CONFIGS; do
      label=${c%%:*}; envs=${c#*:}
      env ${envs//;/ } timeout 400 python bench.py --no-cpu-baseline --no-profile --no-also --steps ${STEPS:-40} --workload $w $ARGS 2>/dev/null | grep '^{' > $out/${label}_${w}_$rep.json
    done
  done
done
python3 - <<PY
import json,glob
cfgs=[c.split(":")[0] for c in "$CONFIGS".split()]
for w in "$WORKLOADS".split():
    vals={c:[json.load(open(f))["value"] for f in sorted(glob.glob("$out/%s_%s_*.json"%(c,w))) if open(f).read().strip()] for c in cfgs}
    base=sum(vals[cfgs[0]])/max(1,len(vals[cfgs[0]]))
    print("%-6s $ARGS " % w + "   ".join("%s %s (%+.1f %%)" % (c, " / ".join("%.1f"%v for v in vals[c]), 100*(sum(vals[c])/max(1,len(vals[c])))/base-100) for c in cfgs))
PY
