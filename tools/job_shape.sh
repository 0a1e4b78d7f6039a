# same-box A/B of the bare MFMA + LDS loop: 32x32x16 (exp2) against 16x16x32 (exp4); wrong results, timing only
# (HIP-event time of the first pass: these builds skip the escalation of the rows they cannot certify)
for r in 1 2; do
for lib in exp2 exp4; do
  export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10 2
  timeout 100 python3 tools/shape_ab.py 250000 1000000 300 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 256 10
done
done
