cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/c1x; mkdir -p $O
for w in c1 c2; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-others > $O/${w}_bench.json 2> $O/ks_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/${w}_kernel_stats.csv; rm -rf $O/ks_$w
  echo "$w: $(python3 tools/show.py $O/${w}_bench.json | cut -c1-120)"
  cut -d, -f1-4 $O/${w}_kernel_stats.csv | head -9 | cut -c1-110
done
