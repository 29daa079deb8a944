timeout 600 python -m pytest tests/test_gpu_post.py -x -q -m gpu 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8
