# round-3 evidence: kernel stats (rocprofv3 --kernel-trace --stats) of every workload next to the bench line of the same run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/prof3; P=$O/profiles; mkdir -p $P
for w in $WL; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $P/r03_${w}_bench_under_rocprof.json 2> $O/ks_$w.err
  echo "rc=$? $w"
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $P/r03_${w}_kernel_stats.csv; rm -rf $O/ks_$w
  python3 tools/show.py $P/r03_${w}_bench_under_rocprof.json | cut -c1-150
  head -8 $P/r03_${w}_kernel_stats.csv | cut -c1-160
done
