# after the new pack / norms kernels and the short sample lists: fuzz the shared sweep, the API and the tiers with new seeds
timeout 1500 python3 tools/fuzz_dual.py 250 201 2>&1 | tail -2
timeout 1500 python3 tools/fuzz_dual.py 20 203 -1 10 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_api.py 200 204 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_tiers.py 120 205 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_longk.py 60 206 2>&1 | tail -1
