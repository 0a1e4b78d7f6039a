timeout 2800 python3 -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | grep -v "^Extension" | tail -3
timeout 2000 python3 tools/fuzz_dual.py 200 1001 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 200 1005 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_api.py 200 1004 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_longk.py 60 1007 2>&1 | tail -1
