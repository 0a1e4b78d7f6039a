timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_merge.py -x -q 2>&1 | grep -v "^Extension" | tail -2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in c3; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v_ks_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > gpurun_out/v_$w.json 2> gpurun_out/v_err.txt
python3 tools/show.py gpurun_out/v_$w.json | cut -c1-140
f=$(find gpurun_out/v_ks_$w -name "*kernel_stats.csv" | head -1); grep -E "mp_empiric" $f | cut -c1-30,100-200; rm -rf gpurun_out/v_ks_$w
done
