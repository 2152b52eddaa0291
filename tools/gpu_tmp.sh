python -m pytest tests/test_vo_gpu.py -x -q 2>&1 | tail -15
