cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=$O/profiles; mkdir -p $P
# default bench line (what the driver runs), un-profiled
timeout 1200 python3 bench.py > $P/r02_bench_default.json 2> $P/r02_bench_default.err
for w in ns c1 c2 c3 c4s; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $P/r02_${w}_bench_under_rocprof.json 2> $O/ks_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $P/r02_${w}_kernel_stats.csv; rm -rf $O/ks_$w
  bash tools/pmc_profile.sh $O/pmc_$w $w
  cp $O/pmc_$w/summary.jsonl $P/r02_${w}_pmc.jsonl
done
ls -la $P
