timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python3 tools/fuzz_tiers.py 100 7 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log; grep BAD $O/fuzz.log | head
for w in ns c4s; do
  timeout 300 python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/b_$w.json 2> $O/b_$w.err; python3 tools/show.py $O/b_$w.json
done
