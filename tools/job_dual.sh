timeout 1500 python3 tools/fuzz_api.py 100 3 > $O/fuzz_api.log 2>&1; grep -c "^ok" $O/fuzz_api.log; grep -B1 "BAD" $O/fuzz_api.log | head -10; grep "^cases\|fault\|Error" $O/fuzz_api.log | head -5
