# idle gaps of the GPU inside a bench step (tools/gaps.py on a kernel trace)     gpurun -- 'WL="ns c3" bash tools/job_gaps.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/gaps; mkdir -p $O
for w in ${WL:-ns c3 c1}; do
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-others --detail bench_detail_$w.json --no-check > $O/$w.json 2> $O/$w.err
  f=$(find $O/kt_$w -name "*kernel_trace.csv" | head -1)
  echo "== $w: $(python3 tools/show.py $O/$w.json | cut -c1-100)"
  python3 tools/gaps.py "$f" ${MIN:-15} $O/${w}_seq.txt | tee $O/${w}_gaps.txt | tail -${TAIL:-45}
  rm -rf $O/kt_$w
done
