cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=$O/profiles; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --workload ns --steps 5 --warmup 2 --no-cpu-baseline --no-others --opt dual_stride=0 > $P/r02_ns_two_searches_bench_under_rocprof.json 2> $O/ks.err
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); cp "$f" $P/r02_ns_two_searches_kernel_stats.csv; rm -rf $O/ks
python3 tools/show.py $P/r02_ns_two_searches_bench_under_rocprof.json | cut -c1-160; python3 tools/ks_show.py $P/r02_ns_two_searches_kernel_stats.csv | head -5
timeout 600 python3 bench.py --workload ns --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check > $O/b.json 2>/dev/null; python3 tools/show.py $O/b.json | cut -c1-120
