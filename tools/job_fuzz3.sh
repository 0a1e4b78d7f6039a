timeout 1500 python3 tools/fuzz_dual.py 300 101 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_dual.py 300 102 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_dual.py 24 103 -1 10 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_api.py 300 104 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 200 105 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_longk.py 200 106 2>&1 | tail -1
