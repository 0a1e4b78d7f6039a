# The committed per-round evidence: default bench line, then per workload the kernel stats (rocprofv3 --kernel-trace --stats) next
# to the bench line of the same run, then the PMC passes (tools/pmc_profile.sh) -> gpurun_out/prof/profiles/ (copy what is to be
# judged into profiles/).     gpurun -- 'ROUND=r04 bash tools/job_prof.sh'     [WL="ns c1 ..."] [PMC_WL="ns c1 ..."]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${ROUND:-r05}
export O=gpurun_out/prof; P=$O/profiles; mkdir -p $P
timeout 1200 python3 bench.py > $P/${R}_bench_default.json 2> $P/${R}_bench_default.err
python3 tools/show.py $P/${R}_bench_default.json | cut -c1-160
for w in ${WL:-ns c1 c2 c3 c4s c4 hard gmm}; do
  steps=5; [ $w = c4 ] && steps=2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline --no-others > $P/${R}_${w}_bench_under_rocprof.json 2> $O/ks_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $P/${R}_${w}_kernel_stats.csv; rm -rf $O/ks_$w
  echo "$w: $(python3 tools/show.py $P/${R}_${w}_bench_under_rocprof.json | cut -c1-120)"
done
for w in ${PMC_WL:-ns c3 c4s c1 c2 hard}; do
  bash tools/pmc_profile.sh $O/pmc_$w $w > /dev/null 2>&1
  cp $O/pmc_$w/summary.jsonl $P/${R}_${w}_pmc.jsonl
  echo "pmc $w: $(wc -l < $P/${R}_${w}_pmc.jsonl) records"
done
# the step's kernels in start order + idle gaps (tools/gaps.py), the Minkowski-family timing beside scikit-learn
for w in ${SEQ_WL:-ns c3}; do
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-others --no-check > /dev/null 2> $O/kt_$w.err
  f=$(find $O/kt_$w -name "*kernel_trace.csv" | head -1)
  python3 tools/gaps.py "$f" 15 $P/${R}_${w}_step_sequence.txt > $P/${R}_${w}_step_gaps.txt; tail -1 $P/${R}_${w}_step_gaps.txt; rm -rf $O/kt_$w
done
timeout 900 python3 tools/minkowski_time.py 15000 300 2>&1 | grep -v amdgpu.ids > $P/${R}_minkowski_time.log; head -3 $P/${R}_minkowski_time.log
ls -la $P
