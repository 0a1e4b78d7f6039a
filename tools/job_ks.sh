cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in ns; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check > $O/${w}.json 2> $O/ks_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/${w}_kernel_stats.csv; rm -rf $O/ks_$w
  echo "== $w: $(python3 tools/show.py $O/$w.json | cut -c1-140)"; python3 tools/ks_show.py $O/${w}_kernel_stats.csv "" | head -28
done
