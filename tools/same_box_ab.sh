#!/bin/bash
# Same MI355X box, same gpurun call, alternating runs: the previous round's tree (built in ./_r03 by `git archive <rev> | tar -x -C _r03` +
# `python -c "from audiossl_amd import build; build.build()"` there) against this tree.  usage (inside gpurun): bash tools/same_box_ab.sh <outdir>
out=${1:-gpurun_out/ab}; mkdir -p $out
for rep in 1 2; do
  for w in clip6 clip2 frame; do
    (cd _r03 && timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 40 --workload $w 2>/dev/null | grep '^{' > ../$out/old_${w}_$rep.json)
    timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 40 --workload $w 2>/dev/null | grep '^{' > $out/new_${w}_$rep.json
  done
done
python - <<PY
import json,glob
for w in ("clip6","clip2","frame"):
    o=[json.load(open(f))["value"] for f in sorted(glob.glob("$out/old_%s_*.json"%w))]
    n=[json.load(open(f))["value"] for f in sorted(glob.glob("$out/new_%s_*.json"%w))]
    print("%-6s previous round %s   this tree %s   change %+.1f %%" % (w, " / ".join("%.1f"%v for v in o), " / ".join("%.1f"%v for v in n), 100*(sum(n)/len(n))/(sum(o)/len(o))-100))
PY
