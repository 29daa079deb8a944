timeout 600 python -m pytest tests/test_map_prep.py tests/test_lpastar.py tests/test_gpu_post.py -x -q -m gpu 2>&1 | grep -a "passed\|failed\|rc=\|Error\|assert\|what" | tail -8
