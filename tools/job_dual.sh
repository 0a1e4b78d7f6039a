timeout 1200 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_sharded_rccl.py -x -q -m gpu > $O/pytest_dual.log 2>&1; tail -25 $O/pytest_dual.log
