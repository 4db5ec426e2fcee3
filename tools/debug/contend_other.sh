pids=""
for i in $(seq 1 8); do taskset -c 0-3 timeout 300 python3 -c "
while True: pass
" & pids="$pids $!"; done
sleep 2
export OMP_NUM_THREADS=4
for wl in clip6 "frame small fp8" "frame base bf16"; do
  (cd _r03 && taskset -c 0-3 timeout 90 python tools/debug/frame_host_profile2.py $wl 2>&1 | grep "host enqueue" | sed 's/^/old tree  /')
  taskset -c 0-3 timeout 90 python tools/debug/frame_host_profile2.py $wl 2>&1 | grep "host enqueue" | sed 's/^/this tree /'
done
kill $pids 2>/dev/null; wait 2>/dev/null
