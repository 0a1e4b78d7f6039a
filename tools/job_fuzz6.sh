timeout 900 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
timeout 2000 python3 tools/fuzz_dual.py 300 601 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_dual.py 24 603 -1 10 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 200 605 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_api.py 200 604 2>&1 | tail -1
