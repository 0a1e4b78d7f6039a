"""profiles/r<NN>_<w>_pmc.jsonl (tools/pmc_profile.sh; the latest round present per workload) -> profiles/pmc_traffic.json: per workload the dominant kernel (largest
total time) with its HBM-side bytes per launch -- FETCH_SIZE x 2 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 B,
MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> B -- clock, matrix-pipe busy and instruction mix.  bench.py reads
`<workload>_<dtype>.hbm_bytes_per_launch` for `roofline.traffic`.      python3 tools/pmc_derive.py [profiles_dir]"""
import collections
import json
import sys
from pathlib import Path

P = Path(sys.argv[1] if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "profiles")
out = {"_note": "HBM-side bytes per launch of the dominant kernel (average over the dispatches of the profiled run): FETCH_SIZE x 2 "
                "(gfx950 correction) + WRITE_SIZE, KiB -> B; separate rocprofv3 --pmc passes (tools/pmc_profile.sh), summaries in "
                "r<NN>_<workload>_pmc.jsonl (latest round per workload; `source` names the file); derived by tools/pmc_derive.py"}
latest = {}
for f in sorted(P.glob("r[0-9][0-9]_*_pmc.jsonl")):
    latest[f.name[4:-len("_pmc.jsonl")]] = f      # sorted: a later round replaces an earlier one
for w, f in sorted(latest.items()):
    per = collections.defaultdict(dict)     # kernel -> counter -> avg ;  plus avg_ns / dispatches
    for line in f.read_text().splitlines():
        r = json.loads(line)
        k = r["kernel"]
        per[k].update(r["counters_avg"])
        if r.get("avg_ns"):
            per[k].setdefault("_avg_ns", r["avg_ns"])
            per[k].setdefault("_disp", r["dispatches"])
    cand = {k: v for k, v in per.items() if "kz_knn_cand" in k and "_avg_ns" in v}
    if not cand:
        continue
    k = max(cand, key=lambda n: cand[n]["_avg_ns"] * cand[n]["_disp"])
    c = cand[k]
    g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                      # per-XCD active cycles
    mf = c.get("SQ_INSTS_MFMA", 0.0) or 1.0
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    tier = "f16" if ("cand_h_" in k or "cand_h64_" in k) else ("bf16x2" if "cand_bf" in k else "f32")
    out[f"{w}_{tier}"] = {
        "source": f.name, "kernel": k.replace("void ", ""), "dispatches_averaged": c["_disp"], "avg_ms_under_pmc": c["_avg_ns"] / 1e6,
        "hbm_bytes_per_launch": (c.get("FETCH_SIZE", 0.0) * 2 + c.get("WRITE_SIZE", 0.0)) * 1024,
        "fetch_size_kib": c.get("FETCH_SIZE"), "write_size_kib": c.get("WRITE_SIZE"),
        "clock_ghz": g / c["_avg_ns"] if g else None,
        "mfma_pipe_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 1024) if g else None,
        "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
        "valu_per_mfma": c.get("SQ_INSTS_VALU", 0.0) / mf, "salu_per_mfma": c.get("SQ_INSTS_SALU", 0.0) / mf,
        "lds_per_mfma": c.get("SQ_INSTS_LDS", 0.0) / mf, "branch_per_mfma": c.get("SQ_INSTS_BRANCH", 0.0) / mf,
        "wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / (c.get("SQ_WAVE_CYCLES") or 1.0),
        "wait_inst_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / (c.get("SQ_WAVE_CYCLES") or 1.0),
        "insts_mfma": c.get("SQ_INSTS_MFMA"),
        "lds_bank_conflict_frac": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / (c.get("SQ_LDS_IDX_ACTIVE") or 1.0),
        "other_kernels": {n.replace("void ", ""): {"dispatches": v["_disp"], "avg_ms": v["_avg_ns"] / 1e6} for n, v in per.items()
                          if n != k and "_avg_ns" in v and v["_avg_ns"] * v["_disp"] > 0.02 * c["_avg_ns"] * c["_disp"]},
    }
(P / "pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps({k: (v if isinstance(v, str) else {kk: v[kk] for kk in ("kernel", "avg_ms_under_pmc", "hbm_bytes_per_launch", "clock_ghz", "mfma_pipe_busy")}) for k, v in out.items() if k != "_note"}, indent=1))
