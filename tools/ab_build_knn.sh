#!/bin/bash
# A/B library with extra flags for kz_knn.hip (finalize / exact kernels, host driver): tools/ab_build_knn.sh <name> <-D flags...>
# -> build/abl/libkiez_amd_<name>.so (every other object taken from the current build)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/abl /tmp/kz_abk_$name
make -s -C kiez_amd/csrc -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" -c kiez_amd/csrc/kz_knn.hip -o /tmp/kz_abk_$name/kz_knn.o
objs=$(ls kiez_amd/csrc/*.o | grep -v "csrc/kz_knn.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/kz_abk_$name/kz_knn.o -o build/abl/libkiez_amd_$name.so
