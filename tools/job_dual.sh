timeout 1200 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 tools/dual_check.py ns c3 2>&1 | tail -3 | cut -c1-300
