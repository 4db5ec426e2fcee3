#!/bin/bash
# Per-kernel averages of the clip2 step with the two-team GEMM on (ATST_TUNE=2001) and off: rocprofv3 kernel stats of both, same box, same call.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in off on; do
  rm -rf /tmp/ttp_$cfg
  if [ $cfg = on ]; then export ATST_TUNE=2001; else unset ATST_TUNE; fi
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ttp_$cfg -o p -- python3 bench.py --workload ${WL:-clip2} --no-cpu-baseline --no-also --no-profile --steps 20 --warmup 5 > /tmp/ttp_$cfg.json 2>/dev/null)
  cut -c1-120 /tmp/ttp_$cfg.json
done
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("/tmp/ttp_off"), load("/tmp/ttp_on")
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0, 0))[2] + b.get(n, (0, 0, 0))[2]))
print("kernel | off: calls avg_us total_ms | on: calls avg_us total_ms")
for n in names[:26]:
    x, y = a.get(n, (0, 0, 0)), b.get(n, (0, 0, 0))
    print(f"{n[:90]:90s} | {x[0]:5d} {x[1]:8.1f} {x[2]:8.1f} | {y[0]:5d} {y[1]:8.1f} {y[2]:8.1f}")
print("total ms:", sum(v[2] for v in a.values()), sum(v[2] for v in b.values()))
PY
