# norms / pack kernel variants: parity of the default build, then per-kernel times from rocprofv3 kernel stats of the same bench command
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dual.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/pack2; mkdir -p $O
for lib in default r1 r2 r8 b16k; do
  if [ $lib = default ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_pk_$lib.so; fi
  for w in c1 ns c3; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check > $O/$w.json 2> $O/$w.err
    f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
    python3 - "$f" $lib $w <<'PY'
import csv,sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'norms' in r['Name'] or 'pack_h' in r['Name']:
        out.append("%s avg %.1f min %.1f max %.1f"%('norms' if 'norms' in r['Name'] else 'pack',float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3))
print(sys.argv[2],sys.argv[3]," | ".join(sorted(out)))
PY
    rm -rf $O/ks
  done
done
