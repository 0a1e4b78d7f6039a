# bench.py --workload $W with several option sets (A/B of internal knobs on one box).
#   gpurun -- 'W=hard SETS="wide_lists=32,wide_sel=256 wide_lists=24,wide_sel=192" bash tools/job_opts.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/opts; mkdir -p $O
for set in ${SETS:-default}; do
  args=""; [ "$set" != default ] && for kv in ${set//,/ }; do args="$args --opt $kv"; done
  timeout 600 python3 bench.py --workload ${W:-ns} --no-others --no-cpu-baseline $args > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  echo "$set: $(python3 tools/show.py $O/b.json | cut -c1-210)"
done
