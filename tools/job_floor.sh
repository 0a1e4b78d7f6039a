#!/bin/bash
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python3 tools/fuzz_dual.py 100 51 > $O/fuzz_dual.log 2>&1; tail -1 $O/fuzz_dual.log
timeout 600 python3 tools/fuzz_tiers.py 60 52 > $O/fuzz_tiers.log 2>&1; tail -1 $O/fuzz_tiers.log
timeout 600 python3 tools/fuzz_longk.py 30 53 > $O/fuzz_longk.log 2>&1; tail -1 $O/fuzz_longk.log
python3 tools/opt_ab.py --workloads c3,ns,c1,c2,c4s --variants "list_floor=1" --rounds 3 --steps 4 --warmup 2 2>&1 | tee $O/ab_now.log
