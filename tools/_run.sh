export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -s -k "fp8 or gemm_tn" 2>&1 | grep -E "dgrad|passed|failed|assert" | head
timeout 300 python bench.py --no-cpu-baseline --steps 20 --workload clip2 --arch base --dtype fp8 --hires 2>/dev/null > gpurun_out/r03_bench_base_fp8_hires.json; cut -c1-120 gpurun_out/r03_bench_base_fp8_hires.json
timeout 300 python bench.py --no-cpu-baseline --steps 20 --workload clip2 --arch base --dtype fp8 2>/dev/null > gpurun_out/r03_bench_base_fp8.json; cut -c1-120 gpurun_out/r03_bench_base_fp8.json
