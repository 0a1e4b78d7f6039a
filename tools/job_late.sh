# LDS-DMA copies issued one slice later (behind the next slice's MFMAs), with the 4-slot and with the 6-slot ring: parity, then A/B
for lib in late latedeep; do
  KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so timeout 900 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_northstar.py -x -q 2>&1 | tail -2
done
export O=gpurun_out/late; mkdir -p $O
WL="ns c4s c1 c3" AB="late latedeep" bash tools/job_ab.sh
