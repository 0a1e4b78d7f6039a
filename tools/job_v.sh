timeout 1800 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python3 tools/fuzz_dual.py 120 41 2>&1 | tail -1
timeout 900 python3 tools/fuzz_tiers.py 80 9 2>&1 | tail -1
timeout 900 python3 tools/fuzz_api.py 60 5 2>&1 | tail -1
