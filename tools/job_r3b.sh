cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dual.py tests/test_gpu_merge.py tests/test_gpu_fullsize.py tests/test_gpu_longk.py -x -q 2>&1 | tail -3
export O=gpurun_out/r3b; mkdir -p $O
for w in c3 ns; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $O/${w}_bench.json 2> $O/ks_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/${w}_kernel_stats.csv; rm -rf $O/ks_$w
  echo "$w: $(python3 tools/show.py $O/${w}_bench.json | cut -c1-150)"
  grep -E "finalize|mp_empiric|scatter|select_kernel" $O/${w}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
