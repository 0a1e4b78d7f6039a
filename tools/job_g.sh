bash tools/pmc_profile.sh $O/pmc_c1 c1
bash tools/pmc_profile.sh $O/pmc_ns ns
grep -h cand_h $O/pmc_c1/summary.jsonl | cut -c1-900
grep -h cand_h $O/pmc_ns/summary.jsonl | cut -c1-900
tail -3 $O/pmc_c1/sq3.log
