export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_precise_gpu.py -q -s 2>&1 | grep -E "precise|passed|failed|Error|assert" | head -40
