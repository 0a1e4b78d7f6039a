# where the ordinary fp16 kernel's time goes on the ns / c4s / C1 shapes: full kernel (stats of the default library with the
# shared sweep off would include finalize; here main_kernel_ms only), exp3 = scan without merges, exp1 = no candidate scan,
# exp2 = bare MFMA + LDS loop (no barrier, no DMA, no scan)
for r in 1 2; do
for lib in default exp3 exp1 exp2; do
  if [ $lib = default ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so; fi
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 300 10
  timeout 100 python3 tools/shape_ab.py 500000 500000 200 50
done
done
