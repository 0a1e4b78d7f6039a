timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_short_lists.py tests/test_gpu_parity.py tests/test_gpu_dual.py -x -q 2>&1 | grep -v "^Extension" | tail -3
timeout 600 python3 tools/short_route_share.py 2>&1 | grep "min_tiles  64"
KZ_METRIC=euclidean timeout 900 python3 tools/short_route_stress.py 400000 200 50 2>&1 | grep "short=1"
timeout 900 python3 tools/short_route_stress.py 400000 200 50 2>&1 | grep "short=1"
