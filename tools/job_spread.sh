KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_spread.so timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q 2>&1 | tail -2
export O=gpurun_out/spread; mkdir -p $O
WL="c4s c3" AB="spread" bash tools/job_ab.sh
