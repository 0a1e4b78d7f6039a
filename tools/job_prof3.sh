# round-3 evidence: default bench line, then per workload kernel stats (rocprofv3 --kernel-trace --stats) next to the bench line
# of the same run, then the PMC passes (tools/pmc_profile.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/prof3; P=$O/profiles; mkdir -p $P
timeout 900 python3 bench.py > $P/r03_bench_default.json 2> $P/r03_bench_default.err
python3 tools/show.py $P/r03_bench_default.json | cut -c1-160
for w in ns c1 c2 c3 c4s; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $P/r03_${w}_bench_under_rocprof.json 2> $O/ks_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $P/r03_${w}_kernel_stats.csv; rm -rf $O/ks_$w
  echo "$w: $(python3 tools/show.py $P/r03_${w}_bench_under_rocprof.json | cut -c1-120)"
done
for w in ns c3 c4s c1 c2; do
  bash tools/pmc_profile.sh $O/pmc_$w $w > /dev/null 2>&1
  cp $O/pmc_$w/summary.jsonl $P/r03_${w}_pmc.jsonl
  echo "pmc $w: $(wc -l < $P/r03_${w}_pmc.jsonl) records"
done
ls -la $P
