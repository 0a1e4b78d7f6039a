# after the new pack / norms kernels, the short sample lists and the short-list route of the main sweep: fuzz with new seeds
timeout 2000 python3 tools/fuzz_dual.py 300 301 2>&1 | tail -4
timeout 1500 python3 tools/fuzz_dual.py 24 303 -1 10 2>&1 | grep -c "^ok"
timeout 1500 python3 tools/fuzz_dual.py 24 303 -1 10 2>&1 | tail -1
