cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 1 2; do
  export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_exp$n.so
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/p$n -- python3 bench.py --workload c1 --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-check > $O/p$n.json 2> $O/p$n.err
  f=$(find $O/p$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" $n <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "cand_h" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    by[r["Dispatch_Id"]]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for d, c in sorted(by.items(), key=lambda kv: int(kv[0]))[-2:]:
    cyc = c["GRBM_GUI_ACTIVE"] / 8; ns = c["_ns"]; wc = c["SQ_WAVE_CYCLES"]
    print("exp", sys.argv[2], "ms %.3f clock %.3f GHz mfma_util %.3f wait_inst %.2f wait_any %.2f active %.2f lds_active/cyc %.2f" % (
        ns / 1e6, cyc / ns, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
        c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)))
PY
  rm -rf $O/p$n
done
