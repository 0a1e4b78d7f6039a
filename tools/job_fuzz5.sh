timeout 1500 python3 tools/fuzz_api.py 200 404 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 150 405 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_dual.py 150 406 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_longk.py 40 407 2>&1 | tail -1
