#!/bin/bash
# Effective clock and MFMA-pipe utilisation of the fused kernel for a given library build (GPU box):
#   tools/pmc_clock.sh OUTDIR [lib.so] [bench args]
# clock = GRBM_GUI_ACTIVE / 8 / kernel time (MI355X_MICROARCH.md, DVFS give-back); util = MFMA_BUSY / (4 SIMD x 256 CU x cycles)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
LIB=${1:-}; shift || true
if [ -n "$LIB" ] && [ "$LIB" != "-" ]; then export KIEZ_AMD_LIB=$PWD/$LIB; fi
mkdir -p "$OUT"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT/run" -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-check "$@" > "$OUT/run.log" 2>&1
f=$(find "$OUT/run" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "cand_bf" in r["Kernel_Name"] or "cand_kernel" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    if "Start_Timestamp" in r:
        by[r["Dispatch_Id"]]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for d, c in sorted(by.items(), key=lambda kv: int(kv[0]))[-3:]:
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    ns = c.get("_ns", 0)
    print("dispatch", d, "ms", round(ns / 1e6, 2), "clock GHz", round(cyc / ns, 3) if ns else None,
          "mfma_util", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 3), "insts_mfma", c["SQ_INSTS_MFMA"],
          "wave_cycles", c["SQ_WAVE_CYCLES"])
PY
grep -o '"achieved": [0-9.]*' "$OUT/run.log" | tail -1
