# bare loop of a one-wave-per-SIMD kernel (exp8: 64 queries per wave, TWICE the work per wave on the same grid) against the
# shipped structure's bare loop (exp2)
for r in 1 2; do
for lib in exp2 exp8; do
  export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 300 10
done
done
