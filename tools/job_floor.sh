#!/bin/bash
python3 tools/opt_ab.py --workloads c3,ns --variants "list_floor=0,floor_probe=1024,floor_margin=1.3;list_floor=1,floor_probe=2048,floor_margin=1.3;list_floor=1,floor_probe=1024,floor_margin=1.3;list_floor=1,floor_probe=1024,floor_margin=1.6;list_floor=1,floor_probe=512,floor_margin=1.6;list_floor=1,floor_probe=512,floor_margin=2.0;list_floor=1,floor_probe=256,floor_margin=2.0" --rounds 3 --steps 4 --warmup 2 2>&1 | tee $O/ab_probe2.log
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python3 tools/fuzz_tiers.py 80 45 > $O/fuzz_tiers.log 2>&1; tail -1 $O/fuzz_tiers.log
timeout 600 python3 tools/fuzz_api.py 40 46 > $O/fuzz_api.log 2>&1; tail -1 $O/fuzz_api.log
timeout 600 python3 tools/fuzz_longk.py 30 47 > $O/fuzz_longk.log 2>&1; tail -1 $O/fuzz_longk.log
