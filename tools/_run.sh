export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm" 2>&1 | tail -3
timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "LN|dgrad" | tee gpurun_out/r03c/gemm_bench.txt
timeout 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/r03c/bench_clip6.json 2> gpurun_out/r03c/bench_clip6.err; cut -c1-260 gpurun_out/r03c/bench_clip6.json
