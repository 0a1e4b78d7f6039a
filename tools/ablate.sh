#!/bin/bash
# Diagnostic libraries of the fp16 fused kernel with parts removed (-DKZ_EXP=n, see kz_knn_h16.h / kz_knn_epi3.h): results
# are WRONG in these builds, they only price the parts.  build/abl/libkiez_amd_exp<n>.so; only the K' = 16 unit is rebuilt.
#   tools/ablate.sh 1 2 3
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl /tmp/kz_abl
make -s -C kiez_amd/csrc -j8
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-honor-nans -DKZ_EXP=$n -c kiez_amd/csrc/kz_knn_h_kp16.hip -o /tmp/kz_abl/h16_$n.o
  # (kz_knn.hip with KZ_EXP: no escalation of the rows these builds cannot certify -- only the first pass runs and is timed)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DKZ_EXP=$n -c kiez_amd/csrc/kz_knn.hip -o /tmp/kz_abl/knn_$n.o
  objs=$(ls kiez_amd/csrc/*.o | grep -v "kz_knn_h_kp16.o\|kz_knn.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/kz_abl/h16_$n.o /tmp/kz_abl/knn_$n.o -o build/abl/libkiez_amd_exp$n.so
done
