#!/bin/bash
# PMC passes for the kernels of one workload (run on the GPU box through gpurun).  Separate rocprofv3 runs per counter
# group, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass;
# --pmc only with --kernel-trace).  Prints per-kernel averages as JSON lines to <outdir>/summary.jsonl.
#   tools/pmc_profile.sh <outdir> <workload> [extra bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; W=$2; shift 2
mkdir -p "$OUT"
ARGS="--workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-others $*"
: > "$OUT/summary.jsonl"
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
  local f=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" >> "$OUT/summary.jsonl" <<'PY'
import csv, sys, collections, json
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
with open(f) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"] == list(agg[k].keys())[0]:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, d in agg.items():
    if "kz_" not in k: continue
    print(json.dumps({"pass": name, "kernel": k[:90], "dispatches": len(next(iter(d.values()))),
                      "avg_ns": (sum(dur[k]) / len(dur[k])) if dur[k] else None,
                      "counters_avg": {c: sum(v) / len(v) for c, v in d.items()}}))
PY
  rm -rf "$OUT/$name"
}
# PASSES="fetch write tcc grbm" selects passes (default: all)
sel() { [ -z "${PASSES:-}" ] || [[ " $PASSES " == *" $1 "* ]]; }
orig_run=$(declare -f run); eval "${orig_run/run ()/run_pass ()}"; run() { sel "$1" && run_pass "$@"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F16
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
