timeout 1500 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_parity.py tests/test_kiez_api.py -x -q -m gpu 2>&1 | tail -4
