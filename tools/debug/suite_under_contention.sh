#!/bin/bash
# the GPU test-suite with the host kept busy: 8 busy loops and the test process pinned to the same 4 cores (everything under `timeout`)
pids=""
for i in $(seq 1 8); do taskset -c 0-3 timeout 1500 python3 -c "
while True: pass
" & pids="$pids $!"; done
sleep 2
taskset -c 0-3 timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
kill $pids 2>/dev/null; wait 2>/dev/null
