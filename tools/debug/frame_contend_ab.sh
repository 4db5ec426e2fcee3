#!/bin/bash
# Host-contention A/B for the ATST-Frame step: the previous tree (_r03 = `git archive <rev> | tar -x -C _r03` + the built library: CPU torch ops in the per-step host
# bookkeeping) against this tree (numpy).  The measured processes and N busy loops are pinned to the same 4 cores; everything runs under `timeout`.
N=${1:-8}
pids=""
for i in $(seq 1 $N); do taskset -c 0-3 timeout 420 python3 -c "
while True: pass
" & pids="$pids $!"; done
sleep 2
export OMP_NUM_THREADS=4
for rep in 1 2; do
  (cd _r03 && taskset -c 0-3 timeout 90 python tools/debug/frame_host_profile2.py frame 2>&1 | grep "host enqueue" | sed 's/^/old tree  /')
  taskset -c 0-3 timeout 90 python tools/debug/frame_host_profile2.py frame 2>&1 | grep "host enqueue" | sed 's/^/this tree /'
done
(cd _r03 && taskset -c 0-3 timeout 100 python bench.py --workload frame --steps 20 --warmup 8 --no-cpu-baseline --no-profile --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old tree   bench frame', d['value'])")
taskset -c 0-3 timeout 100 python bench.py --workload frame --steps 20 --warmup 8 --no-cpu-baseline --no-profile --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('this tree  bench frame', d['value'])"
kill $pids 2>/dev/null
wait 2>/dev/null
