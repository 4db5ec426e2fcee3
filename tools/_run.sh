export TMPDIR=/tmp
for i in 1 2; do
  (cd _r02 && timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | cut -c1-100 | sed 's/^/r02 /')
  timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | cut -c1-100 | sed 's/^/r03 /'
done
(cd _r02 && timeout 300 python bench.py --no-cpu-baseline --steps 30 --workload clip2 2>/dev/null | cut -c1-100 | sed 's/^/r02 clip2 /')
timeout 300 python bench.py --no-cpu-baseline --steps 30 --workload clip2 2>/dev/null | cut -c1-100 | sed 's/^/r03 clip2 /'
(cd _r02 && timeout 300 python bench.py --no-cpu-baseline --steps 30 --workload frame 2>/dev/null | cut -c1-100 | sed 's/^/r02 frame /')
timeout 300 python bench.py --no-cpu-baseline --steps 30 --workload frame 2>/dev/null | cut -c1-100 | sed 's/^/r03 frame /')
(cd _r02 && timeout 300 python bench.py --no-cpu-baseline --steps 20 --workload clip2 --arch base --dtype fp8 2>/dev/null | cut -c1-100 | sed 's/^/r02 base fp8 /')
timeout 300 python bench.py --no-cpu-baseline --steps 20 --workload clip2 --arch base --dtype fp8 2>/dev/null | cut -c1-100 | sed 's/^/r03 base fp8 /'
