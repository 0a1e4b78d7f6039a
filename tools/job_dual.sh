timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu -k duplicates 2>&1 | grep -E "Mismatch|differ|err_msg|dist|ind|Max|x:|y:" | head -20
