cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_e7.so
timeout 600 python3 tools/dual_check.py ns 2>&1 | tail -3 | head -1 | cut -c1-200
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_n -- python3 tools/dual_check.py ns > $O/pmc_n.log 2>&1
f=$(find $O/pmc_n -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "cand_h" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    h, m = sum(d["TCC_HIT_sum"]), sum(d["TCC_MISS_sum"])
    print(k[-40:], "dispatches", len(d["TCC_HIT_sum"]), "L2 hit %.3f" % (h / (h + m)))
PY
rm -rf $O/pmc_n
