# The raw evidence behind "the headline sweep sits on the chip's power limit" (HISTORY.md 7c):
#   1. tools/power_probe.py: the SAME launch on three operand kinds (uniform random / small integers / constant rows = an all-zero
#      centred fp16 image), ns-like shape and C1 -> <P>/r05_power_probe.log
#   2. the same program under rocprofv3 --pmc (separate passes; program directly behind `--`): GRBM_GUI_ACTIVE (clock the chip held =
#      GRBM / 8 XCDs / duration) and SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy) per dispatch of the fused kernel
#      -> <P>/r05_power_probe.pmc.jsonl (one record per dispatch of the sweep, in launch order: 3 operand kinds x 3 launches x 2 shapes)
#   3. tools/sweep_intercept.py (C1's start burst) -> <P>/r05_sweep_intercept.log
#      gpurun -- 'bash tools/job_power.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/power; P=$O/profiles; mkdir -p $P
timeout 900 python3 tools/power_probe.py > $P/r05_power_probe.log 2>&1
cat $P/r05_power_probe.log
: > $P/r05_power_probe.pmc.jsonl
for pass in "grbm GRBM_GUI_ACTIVE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA"; do
  set -- $pass; name=$1; shift
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 tools/power_probe.py 3 > $O/$name.log 2>&1
  f=$(find $O/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" >> $P/r05_power_probe.pmc.jsonl <<'PY'
import csv, sys, json, collections
f, name = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
with open(f) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0]
        if "kz_knn_cand_h" not in k:
            continue
        key = (int(r["Dispatch_Id"]), k)
        rec = rows.setdefault(key, {"start": float(r["Start_Timestamp"]), "end": float(r["End_Timestamp"]), "c": {}})
        rec["c"][r["Counter_Name"]] = rec["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for n, ((did, k), rec) in enumerate(sorted(rows.items(), key=lambda kv: kv[1]["start"])):
    ns = rec["end"] - rec["start"]
    if ns < 1e6:      # (re-searches of a handful of rows: not the sweep)
        continue
    out = {"pass": name, "order": n, "dispatch": did, "kernel": k[:80], "ms": ns / 1e6, "counters": rec["c"]}
    if "GRBM_GUI_ACTIVE" in rec["c"]:
        out["clock_ghz"] = rec["c"]["GRBM_GUI_ACTIVE"] / 8.0 / ns
    print(json.dumps(out))
PY
  rm -rf $O/$name
done
wc -l $P/r05_power_probe.pmc.jsonl
[ -n "${SKIP_INTERCEPT:-}" ] || { timeout 600 python3 tools/sweep_intercept.py > $P/r05_sweep_intercept.log 2>&1; cat $P/r05_sweep_intercept.log; }
