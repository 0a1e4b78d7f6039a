timeout 900 python3 -m pytest tests -m gpu -x -q -k "parity" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python3 tools/fuzz_tiers.py 80 11 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep BAD $O/fuzz.log | head
for w in ns c3 c1; do
  timeout 300 python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-others > $O/b_$w.json 2> $O/b_$w.err; python3 tools/show.py $O/b_$w.json
  timeout 300 python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-others --opt h_wps=2 > $O/b_${w}_w2.json 2> $O/b_${w}_w2.err; python3 tools/show.py $O/b_${w}_w2.json
done
