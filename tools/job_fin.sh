# finalize kernel variants (rows per group / groups in flight / minimum waves per SIMD): finalize_avg of ns and c3, second stream off
mkdir -p gpurun_out/fin
for r in 1 2; do
for lib in r4d2w4 r2d3w5 r2d4w5 r1d4w6 r2d2w6 r1d3w7 r1d3w6 r1d2w7 r3d2w5; do
  export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_fin_$lib.so
  for wl in ns c3; do
  timeout 300 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check --opt dual_overlap=0 > gpurun_out/fin/${wl}_${lib}_$r.json 2> gpurun_out/fin/err.txt
  python3 - $wl $lib $r <<'PY'
import json,sys
wl,lib,r=sys.argv[1:4]
j=json.loads(open(f'gpurun_out/fin/{wl}_{lib}_{r}.json').read().strip().splitlines()[-1])
print(f"{lib} r{r} {wl}: ms/step {j['ms_per_step']:.2f} main {j['roofline']['avg_launch_ms']:.2f} fin_avg {j['other_kernels_ms']['finalize_avg']:.3f} rev_extra {j['shared_sweep']['reverse_extra_ms_per_step']:.2f}")
PY
  done
done
done
