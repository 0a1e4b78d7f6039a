for r in 1 2; do
for lib in exp1 exp7 exp5; do
  export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
done
done
