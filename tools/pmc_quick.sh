#!/bin/bash
# quick TCC hit/miss + FETCH_SIZE for given bench args
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
mkdir -p "$OUT"
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1
  f=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "cand" in k: print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
  grep -o '"achieved": [0-9.]*' "$OUT/$name.log" | tail -1
done
