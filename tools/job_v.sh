cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in c3 ns; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v_ks_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > gpurun_out/v_$w.json 2> gpurun_out/v_err.txt
python3 tools/show.py gpurun_out/v_$w.json | cut -c1-200
f=$(find gpurun_out/v_ks_$w -name "*kernel_stats.csv" | head -1); grep -E "theta|select_kernel|scatter" $f | cut -c1-60,160-260; rm -rf gpurun_out/v_ks_$w
done
