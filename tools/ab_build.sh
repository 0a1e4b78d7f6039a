#!/bin/bash
# A/B library for same-box comparisons: tools/ab_build.sh <name> <extra hipcc flags for the fp16 units...>
# -> build/abl/libkiez_amd_<name>.so (everything else taken from the current objects)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/abl /tmp/kz_ab_$name
make -s -C kiez_amd/csrc -j8
objs=$(ls kiez_amd/csrc/*.o | grep -v "kz_knn_h_kp\|kz_knn_hd_kp\|kz_knn_h64")
hobjs=""
for kp in 16 32 64 128; do for v in h hd; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-honor-nans -Wno-inline-asm "$@" -c kiez_amd/csrc/kz_knn_${v}_kp$kp.hip -o /tmp/kz_ab_$name/$v$kp.o &
  hobjs="$hobjs /tmp/kz_ab_$name/$v$kp.o"
done; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-honor-nans -Wno-inline-asm "$@" -c kiez_amd/csrc/kz_knn_h64.hip -o /tmp/kz_ab_$name/q64.o &
hobjs="$hobjs /tmp/kz_ab_$name/q64.o"
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $hobjs -o build/abl/libkiez_amd_$name.so
