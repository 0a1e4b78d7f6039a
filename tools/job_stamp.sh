export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_stamp.so
timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10 2>&1 | grep -v "^$" | tail -6
timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10 2 2>&1 | tail -3
timeout 100 python3 tools/shape_ab.py 100000 100000 128 10 2>&1 | tail -3
timeout 100 python3 tools/shape_ab.py 500000 500000 200 50 2>&1 | tail -3
