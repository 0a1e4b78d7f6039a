timeout 2500 python3 tools/fuzz_dual.py 40 1403 -1 10 2>&1 | tail -1
timeout 2000 python3 tools/fuzz_api.py 300 1404 2>&1 | tail -1
timeout 2000 python3 tools/fuzz_dual.py 300 1401 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 300 1405 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_longk.py 80 1407 2>&1 | tail -1
