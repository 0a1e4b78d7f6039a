timeout 1500 python3 tools/fuzz_longk.py 80 1 2>&1 | tail -25
timeout 900 python3 tools/fuzz_api.py 120 31 2>&1 | tail -2
timeout 900 python3 tools/fuzz_tiers.py 80 7 2>&1 | tail -2
