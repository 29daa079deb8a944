timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -a "passed\|failed\|Error\|assert\|seed" | tail -8
