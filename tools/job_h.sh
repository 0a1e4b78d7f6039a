cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 0 1 2 3; do
  for w in c1 ns; do
    if [ $n = 0 ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_exp$n.so; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_${n}_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-check > $O/e${n}_$w.json 2> $O/e${n}_$w.err
    f=$(find $O/ks_${n}_$w -name "*kernel_stats.csv" | head -1)
    cp "$f" $O/ks_${n}_$w.csv; echo "exp $n $w:"; python3 tools/ks_show.py "$f" cand_h_kernel
    rm -rf $O/ks_${n}_$w
  done
done
