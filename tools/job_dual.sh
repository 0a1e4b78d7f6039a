timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu 2>&1 | tail -2
timeout 1500 python3 tools/fuzz_dual.py 200 11 > $O/fuzz2.log 2>&1; grep -c "^ok" $O/fuzz2.log; grep "BAD\|^cases\|fault" $O/fuzz2.log | head -5
timeout 900 python3 tools/fuzz_tiers.py 80 > $O/fuzz3.log 2>&1; tail -1 $O/fuzz3.log
