# merged kernel + HIP API timeline of one steady-state bench step (tools/step_trace.py)   gpurun -- 'WL="ea15k c2" bash tools/job_steptrace.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/steptrace; mkdir -p $O
for w in ${WL:-ea15k}; do
  timeout 400 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-others --no-check --detail bench_detail_$w.json ${BARGS:-} > $O/$w.json 2> $O/$w.err
  k=$(find $O/kt_$w -name "*kernel_trace.csv" | head -1); h=$(find $O/kt_$w -name "*hip_api_trace.csv" | head -1)
  python3 tools/step_trace.py "$k" "$h" ${MIN:-20} > $O/${w}_step.txt; tail -${TAIL:-20} $O/${w}_step.txt
  rm -rf $O/kt_$w
done
