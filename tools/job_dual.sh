timeout 1700 python3 tools/fuzz_dual.py 300 101 > $O/fuzz_a.log 2>&1; grep -c "^ok" $O/fuzz_a.log; grep "BAD\|^cases\|fault" $O/fuzz_a.log | head -5
timeout 900 python3 tools/fuzz_api.py 150 202 > $O/fuzz_b.log 2>&1; grep -c "^ok" $O/fuzz_b.log; grep -B1 "BAD" $O/fuzz_b.log | head -6; grep "^cases\|fault" $O/fuzz_b.log | head -3
timeout 900 python3 tools/fuzz_tiers.py 150 > $O/fuzz_c.log 2>&1; tail -1 $O/fuzz_c.log
