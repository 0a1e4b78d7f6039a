#!/bin/bash
python3 -m pytest tests/test_gpu_short_lists.py tests/test_gpu_floor.py tests/test_gpu_dual.py -x -q 2>&1 | tail -3
python3 tools/opt_ab.py --workloads c3,ns,hard --variants "list_floor=0;list_floor=1" --rounds 3 --steps 4 --warmup 2 --check 2>&1 | tee $O/ab_sub.log
