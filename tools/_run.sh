export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | grep -E "assert|Error|rel|passed|failed" | head -30
