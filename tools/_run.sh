export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_step_gpu.py -x -q -s -k "mel or hires or depth2 or encoder_gradient" 2>&1 | grep -E "hires|passed|failed|Error|assert" | head -30
