# what the list events cost: the same sweep (a) as built, (b) again from its own final thresholds ("abl" = 1: K' insertions per
# query instead of K' (1 + ln(n / K'))), (c) with no event ever logged (build/abl/libkiez_amd_noev.so: tools/ab_build.sh noev -DKZ_ABL_NO_EVENTS)
#   gpurun -- 'bash tools/job_event_ablation.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/event_ablation; mkdir -p $O
for shape in "100000 100000 128 10" "98304 409600 128 10" "100000 100000 200 10" "250000 1000000 200 10"; do
  echo "== $shape"
  python3 tools/shape_ab.py $shape tier_probe=0 2>&1 | tail -1
  python3 tools/shape_ab.py $shape tier_probe=0 abl=1 2>&1 | tail -1
  KIEZ_AMD_LIB=build/abl/libkiez_amd_noev.so python3 tools/shape_ab.py $shape tier_probe=0 2>&1 | tail -1
done | tee $O/ablation.log
