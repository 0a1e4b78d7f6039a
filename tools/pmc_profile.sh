#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box through gpurun).  Separate rocprofv3 runs per counter
# group, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   tools/pmc_profile.sh <outdir> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
mkdir -p "$OUT"
ARGS="${@:---steps 2 --warmup 1 --no-cpu-baseline}"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
  local f=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
  echo "== $name: $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(f) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "kz_" not in k: continue
    print(k, {c: (sum(v)/len(v), len(v)) for c, v in d.items()})
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
