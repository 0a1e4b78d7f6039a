"""GPU box: same-process, interleaved A/B of context options on bench workloads (guide rule 24: variants x rounds in ONE process).

    python3 tools/opt_ab.py --workloads ns,c1 --variants "pack_sweep=0;pack_sweep=1" [--rounds 3] [--steps 5] [--warmup 2]

A variant is a comma-separated list of name=value context options; options stay set, so every variant names every option
that any variant changes.  Prints per workload and variant the step
time and the main-kernel time of every round, their medians and minima, and the escalated / fallback row counts."""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("KIEZ_AMD_WITH_TORCH", "1")
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="ns")
    ap.add_argument("--variants", default=";")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from kiez_amd.distributed import Comm, HipEngine
    eng = HipEngine(0)
    comm = Comm()
    variants = [v.strip() for v in args.variants.split(";")]
    out = {}
    for w in args.workloads.split(","):
        rec = {v: {"step": [], "main": [], "esc": [], "resc": [], "fb": [], "check": None} for v in variants}
        for r in range(args.rounds):
            for v in variants:
                opts = dict(o.split("=") for o in filter(None, v.split(",")))
                for name, val in opts.items():
                    eng.ctx.set_option(name, float(val))
                eng.ctx.trim()
                s, _, _ = bench.run_workload(w, eng, comm, dist, 0, 1, args.steps, args.warmup, check=args.check and r == 0, check_rows=256)
                rec[v]["step"].append(s["ms_per_step"])
                rec[v]["main"].append(s["kernel_s"] / max(s["n_launch"], 1) * 1e3)
                rec[v]["esc"].append(s["escalated_rows"])
                rec[v]["resc"].append(s.get("reverse_escalated_rows", 0))
                rec[v]["fb"].append(s["fallback_rows"])
                if s["check"] is not None:
                    rec[v]["check"] = s["check"]
        for v in variants:
            st, mn = sorted(rec[v]["step"]), sorted(rec[v]["main"])
            print(f"{w:5s} [{v or 'defaults':40s}] step med {st[len(st) // 2]:8.3f} min {st[0]:8.3f} | main med {mn[len(mn) // 2]:8.3f} min {mn[0]:8.3f}"
                  f" | esc {rec[v]['esc'][-1]} rev-esc {rec[v]['resc'][-1]} fb {rec[v]['fb'][-1]} | rounds step {[round(x, 2) for x in rec[v]['step']]} main {[round(x, 2) for x in rec[v]['main']]}"
                  + (f" | check {rec[v]['check']}" if rec[v]["check"] else ""), flush=True)
        out[w] = rec
    if args.json:
        Path(args.json).write_text(json.dumps(out))


if __name__ == "__main__":
    main()
