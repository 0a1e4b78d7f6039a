timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python3 tools/fuzz_tiers.py 80 41 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep BAD $O/fuzz.log | head
for w in c3 ns c1; do
  timeout 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $O/b_$w.json 2> $O/b_$w.err; python3 tools/show.py $O/b_$w.json
done
