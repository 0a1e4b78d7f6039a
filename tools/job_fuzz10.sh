timeout 2500 python3 tools/fuzz_dual.py 600 3001 2>&1 | tail -1
timeout 2500 python3 tools/fuzz_api.py 500 3004 2>&1 | tail -1
timeout 2500 python3 tools/fuzz_tiers.py 600 3005 2>&1 | tail -1
timeout 2500 python3 tools/fuzz_longk.py 300 3007 2>&1 | tail -1
timeout 2500 python3 tools/fuzz_dual.py 40 3003 -1 10 2>&1 | tail -1
