#!/bin/bash
# Build ablated variants of the fused kernel (diagnostic; results are wrong by construction) and time them.
# Run here to build:   tools/ablate.sh build      -> build/abl/libkiez_amd_ablN.so
# Run on the GPU box:  tools/ablate.sh run [bench args]
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
if [ "$1" = "build" ]; then
  for n in ${ABL_BUILD:-1 2 3 4 5}; do
    for f in kz_runtime kz_pack kz_knn kz_knn_bf_kp16 kz_knn_bf_kp32 kz_knn_bf_kp64 kz_knn_bf_kp128 kz_hubness kz_analysis; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DKZ_ABLATE=$n -c kiez_amd/csrc/$f.hip -o /tmp/abl_$f.o 2>/dev/null
    done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/abl_kz_runtime.o /tmp/abl_kz_pack.o /tmp/abl_kz_knn.o /tmp/abl_kz_knn_bf_kp16.o /tmp/abl_kz_knn_bf_kp32.o /tmp/abl_kz_knn_bf_kp64.o /tmp/abl_kz_knn_bf_kp128.o /tmp/abl_kz_hubness.o /tmp/abl_kz_analysis.o -o build/abl/libkiez_amd_abl$n.so
  done
  ls -la build/abl
else
  shift || true
  for n in ${ABL_SET:-0 1 2 3 4 5}; do
    if [ $n = 0 ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_abl$n.so; fi
    python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $n', round(d['roofline']['achieved'],1), 'TF', round(d['roofline']['avg_launch_ms'],2), 'ms')"
  done
fi
