#!/bin/bash
# Host synchronisations per training step: rocprofv3 --hip-trace of bench.py at two step counts, per-API call-count difference / 10.
# usage (GPU box): bash tools/sync_count.sh [clip6|clip2|frame] [extra bench args...]
export TMPDIR=/tmp
wl=${1:-clip6}; shift
out=gpurun_out/hiptrace_$wl; rm -rf $out; mkdir -p $out
for n in 5 15; do
  timeout 300 rocprofv3 --hip-trace --output-format csv -d $out/s$n -o t -- python3 bench.py --workload $wl --steps $n --warmup 2 --no-cpu-baseline --no-profile "$@" > /dev/null 2> $out/err$n.txt
done
python3 - "$out" <<'PY'
import collections, csv, glob, sys
def count(d):
    c = collections.Counter()
    for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c[r.get("Function") or r.get("Name") or r.get("Operation")] += 1
    return c
a, b = count(sys.argv[1] + "/s5"), count(sys.argv[1] + "/s15")
print("HIP API calls per training step ((15-step run - 5-step run) / 10):")
for k in sorted(set(a) | set(b), key=lambda k: -(b[k] - a[k])):
    per = (b[k] - a[k]) / 10.0
    if per or any(s in k for s in ("Synchronize", "Memcpy", "EventQuery")):
        print(f"  {k:44s} {per:8.1f}")
PY
rm -rf $out/s5 $out/s15
