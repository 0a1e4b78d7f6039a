#!/bin/bash
# Diagnostic library with in-kernel s_memtime stamps (-DKZ_STAMP): build/abl/libkiez_amd_stamp.so.  Never timed or shipped.
#   tools/build_stamp.sh [extra hipcc flags]      (KZ_STAMP_OUT=name overrides the output file name)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl /tmp/kz_stamp_obj
objs=""
pids=""
for f in $(make -s -C kiez_amd/csrc print-srcs); do
  b=${f%.hip}
  fl=""; case $f in kz_knn_h_kp*|kz_knn_hd_kp*) fl="-fno-honor-nans";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DKZ_STAMP $fl "$@" -c kiez_amd/csrc/$f -o /tmp/kz_stamp_obj/$b.o &
  pids="$pids $!"
  objs="$objs /tmp/kz_stamp_obj/$b.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o build/abl/${KZ_STAMP_OUT:-libkiez_amd_stamp.so}
