# bare loop with 64 queries per wave at TWO waves per SIMD (exp9, d <= 128 only: 230 VGPRs) against the shipped structure's (exp2)
for r in 1 2; do
for lib in exp2 exp9; do
  export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10
  timeout 100 python3 tools/shape_ab.py 400000 400000 128 10
  timeout 100 python3 tools/shape_ab.py 200000 200000 64 10
done
done
