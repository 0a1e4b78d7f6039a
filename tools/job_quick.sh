# Quick GPU cycle: selected tests, then single-workload bench lines.   gpurun -- 'T="tests/a.py tests/b.py" W="ns c3" bash tools/job_quick.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${O:-gpurun_out/quick}; mkdir -p $O
if [ -n "${T:-}" ]; then
  timeout ${TT:-1500} python3 -m pytest $T -x -q > $O/pytest.log 2>&1; tail -${TAIL:-15} $O/pytest.log
fi
for w in ${W:-}; do
  timeout 600 python3 bench.py --workload $w --no-others --detail bench_detail_$w.json --no-cpu-baseline ${BARGS:-} > $O/bench_$w.json 2> $O/bench_$w.err || tail -5 $O/bench_$w.err
  echo "$w: $(python3 tools/show.py $O/bench_$w.json | cut -c1-330)"
done
