timeout 2800 python3 -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | grep -v "^Extension" | tail -3
timeout 2000 python3 tools/fuzz_dual.py 150 1101 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 250 1105 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_api.py 120 1104 2>&1 | tail -1
