# Kernel stats of single workloads (rocprofv3 --kernel-trace --stats; the program directly behind `--`).
#   gpurun -- 'WL="hard ns" [BARGS="--opt x=y"] bash tools/job_ks.sh'   -> gpurun_out/ks/<w>_kernel_stats.csv + bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ks; mkdir -p $O
for w in ${WL:-ns}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline --no-others --no-check ${BARGS:-} > $O/${w}_bench.json 2> $O/${w}.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/${w}_kernel_stats.csv; rm -rf $O/ks_$w
  echo "== $w: $(python3 tools/show.py $O/${w}_bench.json | cut -c1-200)"
  python3 tools/ks_show.py $O/${w}_kernel_stats.csv kz_ 2>/dev/null | head -${ROWS:-16}
done
