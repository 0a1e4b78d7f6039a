#!/bin/bash
# Diagnostic library with in-kernel s_memtime stamps (-DKZ_STAMP): build/abl/libkiez_amd_stamp.so.  Never timed or shipped.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
objs=""
pids=""
for f in kz_runtime kz_pack kz_knn kz_knn_bf_kp16 kz_knn_bf_kp32 kz_knn_bf_kp64 kz_knn_bf_kp128 kz_hubness kz_analysis; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DKZ_STAMP "$@" -c kiez_amd/csrc/$f.hip -o /tmp/st_$f.o &
  pids="$pids $!"
  objs="$objs /tmp/st_$f.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o build/abl/libkiez_amd_stamp.so
