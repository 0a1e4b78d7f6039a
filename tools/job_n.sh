timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for w in c2 ns; do
  timeout 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $O/b_$w.json 2> $O/b_$w.err; python3 tools/show.py $O/b_$w.json
done
