# finalize gather variants once more, on C3's lists as they are now (160 / 128 entries, ~52 gathered rows per query)
for r in 1 2; do
for lib in default r1d3w7 r2d2w6 r2d2w7 r1d3w6; do
  if [ $lib = default ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_fin_$lib.so; fi
  for wl in c3 ns; do
  timeout 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check > gpurun_out/v_tmp.json 2> gpurun_out/v_err.txt
  echo "$lib $wl $(python3 tools/show.py gpurun_out/v_tmp.json | cut -c12-125)"
  done
done
done
