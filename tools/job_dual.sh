timeout 600 python3 tools/_k1.py 2>&1 | tail -6
timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python3 tools/dual_check.py ns 2>&1 | tail -2 | head -1 | cut -c1-260
