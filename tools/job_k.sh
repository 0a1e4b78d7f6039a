cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -m gpu -x -q -k "sort or golden or oracle_parity" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for w in c3 c2 ns; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-others > $O/b_$w.json 2> $O/b_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/ks_$w.csv; rm -rf $O/ks_$w
  echo "== $w: $(python3 tools/show.py $O/b_$w.json)"; python3 tools/ks_show.py $O/ks_$w.csv "" | head -14
done
