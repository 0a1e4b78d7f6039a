#!/bin/bash
# fuzz campaign: gpurun -- 'bash tools/r2_run.sh fuzz tools/job_fuzz.sh'   (SEED0 shifts the seeds)
S=${SEED0:-100}
timeout 900 python3 tools/fuzz_dual.py 400 $((S+1)) > $O/fuzz_dual.log 2>&1; tail -1 $O/fuzz_dual.log
timeout 900 python3 tools/fuzz_dual.py 30 $((S+2)) -1 8 > $O/fuzz_dual_big.log 2>&1; tail -1 $O/fuzz_dual_big.log
timeout 900 python3 tools/fuzz_tiers.py 400 $((S+3)) > $O/fuzz_tiers.log 2>&1; tail -1 $O/fuzz_tiers.log
timeout 600 python3 tools/fuzz_api.py 100 $((S+4)) > $O/fuzz_api.log 2>&1; tail -1 $O/fuzz_api.log
timeout 600 python3 tools/fuzz_longk.py 60 $((S+5)) > $O/fuzz_longk.log 2>&1; tail -1 $O/fuzz_longk.log
