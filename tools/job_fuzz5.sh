# Round-5 fuzz campaign (fixed seeds; logs -> gpurun_out/fuzz5/):   gpurun -- 'SEED0=700 bash tools/job_fuzz5.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fuzz5; mkdir -p $O
S=${SEED0:-700}
timeout 1200 python3 tools/fuzz_dual.py ${N_DUAL:-250} $((S+1)) > $O/fuzz_dual.log 2>&1; tail -1 $O/fuzz_dual.log
timeout 900 python3 tools/fuzz_dual.py ${N_BIG:-20} $((S+2)) -1 8 > $O/fuzz_dual_big.log 2>&1; tail -1 $O/fuzz_dual_big.log
timeout 1200 python3 tools/fuzz_tiers.py ${N_TIERS:-300} $((S+3)) > $O/fuzz_tiers.log 2>&1; tail -1 $O/fuzz_tiers.log
timeout 600 python3 tools/fuzz_api.py ${N_API:-80} $((S+4)) > $O/fuzz_api.log 2>&1; tail -1 $O/fuzz_api.log
timeout 900 python3 tools/fuzz_longk.py ${N_LONGK:-60} $((S+5)) > $O/fuzz_longk.log 2>&1; tail -1 $O/fuzz_longk.log
timeout 900 python3 tools/fuzz_family.py ${N_FAMILY:-100} $((S+6)) > $O/fuzz_family.log 2>&1; tail -1 $O/fuzz_family.log
grep -h "^BAD" $O/*.log | head -20
