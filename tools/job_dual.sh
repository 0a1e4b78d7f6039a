timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu 2>&1 | tail -2
timeout 1500 python3 tools/fuzz_dual.py 100 303 > $O/fuzz_a.log 2>&1; grep -c "^ok" $O/fuzz_a.log; grep "BAD\|^cases\|fault" $O/fuzz_a.log | head -3
timeout 300 python3 tools/dual_check.py ns 2>&1 | tail -2 | head -1 | cut -c1-220
