# deep (6-slot) ring at three workgroups per CU: correctness first, then same-box A/B against the 4-slot build
timeout 900 python3 -m pytest tests/test_gpu_northstar.py tests/test_gpu_dual.py -x -q 2>&1 | tail -5
timeout 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "long_sweeps or two_independent or long_index" 2>&1 | tail -3
export O=gpurun_out/deep; mkdir -p $O
WL="ns c4s" AB="nodeep" bash tools/job_ab.sh
for lib in default nodeep; do
  if [ $lib = default ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so; fi
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
  timeout 100 python3 tools/shape_ab.py 300000 300000 144 10
done
