timeout 900 python3 -m pytest tests -m gpu -x -q -k "parity or fullsize" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python3 tools/fuzz_tiers.py 60 31 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep BAD $O/fuzz.log | head
AB=interleaved WL="c3 c1 ns" bash tools/job_ab.sh
