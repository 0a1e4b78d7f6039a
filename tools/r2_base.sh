set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2base; mkdir -p $O
for w in c1 c2 c3 ns; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-check > $O/ks_$w.json 2> $O/ks_$w.err
  f=$(find $O/ks_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/ks_${w}_kernel_stats.csv 2>/dev/null
  rm -rf $O/ks_$w
done
for w in c1 ns c3; do
  KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_stamp.so python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-check > $O/stamp_$w.json 2> $O/stamp_$w.err
done
tail -3 $O/stamp_*.err
