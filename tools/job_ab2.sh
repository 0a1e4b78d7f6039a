export AB=nopin WL="c1 ns"
bash tools/job_ab.sh
for lib in default nopin; do
  if [ $lib = default ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so; fi
  timeout 600 python3 tools/dual_check.py ns 2>&1 | tail -2 | head -1 | cut -c1-330
done
