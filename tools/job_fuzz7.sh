timeout 2000 python3 tools/fuzz_dual.py 250 901 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 250 905 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_api.py 250 904 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_longk.py 60 907 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_dual.py 16 903 -1 10 2>&1 | tail -1
