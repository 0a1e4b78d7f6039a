for w in c1 ns c4s c3; do
  timeout 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $O/b_$w.json 2> $O/b_$w.err; python3 tools/show.py $O/b_$w.json
done
