KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_ring8.so timeout 900 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_northstar.py -x -q 2>&1 | tail -2
export O=gpurun_out/ring8; mkdir -p $O
WL="ns" AB="ring8" bash tools/job_ab.sh
for lib in default ring8; do
  if [ $lib = default ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so; fi
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
  timeout 100 python3 tools/shape_ab.py 300000 300000 144 10
done
