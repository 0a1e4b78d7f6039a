timeout 900 python3 tools/fuzz_dual.py 120 11 2>&1 | tail -4
timeout 900 python3 tools/fuzz_dual.py 120 12 2>&1 | tail -3
timeout 600 python3 tools/fuzz_dual.py 10 13 -1 10 2>&1 | tail -3
timeout 900 python3 tools/fuzz_api.py 80 21 2>&1 | tail -3
timeout 900 python3 tools/fuzz_tiers.py 2>&1 | tail -3
