# what the DMA ring costs the scan-less kernel: exp1 = ring + barriers, exp5 = barriers only, exp6 = DMA only, exp2 = neither
for r in 1 2; do
for lib in exp1 exp5 exp6 exp2; do
  export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10 2
done
done
