timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu 2>&1 | tail -2
