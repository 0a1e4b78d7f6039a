#!/bin/bash
python3 tools/opt_ab.py --workloads c4s --variants "list_floor=0;list_floor=1,floor_margin=1.3;list_floor=1,floor_margin=2.0;list_floor=1,floor_margin=3.0" --rounds 3 --steps 4 --warmup 2 2>&1 | tee $O/ab_c4s.log
python3 tools/opt_ab.py --workloads c1,c2 --variants "tier_probe=4096,probe_min_pairs=5e10;tier_probe=512,probe_min_pairs=1e9;tier_probe=256,probe_min_pairs=1e9;tier_probe=1024,probe_min_pairs=1e9" --rounds 3 --steps 20 --warmup 3 --check 2>&1 | tee $O/ab_c1.log
