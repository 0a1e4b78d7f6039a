#!/usr/bin/env python3
"""Randomised cross-check of the precision tiers on the GPU box: for random shapes / dtypes / metrics / k the split-bf16
default, the float32-operand kernel and (for small cases) the oracle must agree bit for bit.

    python3 tools/fuzz_tiers.py [n_cases] [seed]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402
from oracle import kiez_oracle as O  # noqa: E402  (checker only)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = N.Context.get()
    bad = 0
    for c in range(n_cases):
        d = int(rng.choice([3, 16, 17, 31, 33, 48, 64, 65, 96, 100, 128, 129, 150, 176, 192, 200, 224, 240, 256, 257, 300, 384, 385, 500]))
        n_t = int(rng.choice([60, 127, 128, 129, 500, 1000, 2049, 4000, 9000, 20000]))
        n_s = int(rng.choice([1, 31, 128, 130, 700, 1500]))
        metric = str(rng.choice(["euclidean", "sqeuclidean", "cosine"]))
        dtype = np.float32 if rng.rand() < 0.6 else np.float64
        k = int(rng.choice([1, 2, 5, 10, 11, 12, 13, 26, 27, 50, 54, 55, 100, 110]))
        single = rng.rand() < 0.25
        k = min(k, n_t - 2, 109 if single else 110)
        gen = rng.randn if rng.rand() < 0.5 else rng.rand
        t = gen(n_t, d).astype(dtype)
        s = t if single else gen(n_s, d).astype(dtype)
        if metric == "cosine":
            s, t = s.astype(np.float64), (s if single else t).astype(np.float64)
        # short-list route of the ordinary kernel forced onto small index ranges; a bound that sends rows down the tiers
        min_tiles, eps = int(rng.choice([64, 2, 2, 4])), float(rng.choice([1.0, 1.0, 1.0, 30.0]))
        ctx.set_option("short_ord_min_tiles", min_tiles)
        ctx.set_option("eps_scale", eps)
        q64 = int(rng.randint(0, 2))          # fp16 pass on the 64-queries-per-wave kernel where it is built (K' = 16, 4 .. 13 slices)
        ctx.set_option("h_q64", q64)
        # tier probe + its ladder (default lists -> wide route -> split-bf16) on these small shapes: probe rows, lists and
        # selected entries of the wide route drawn at random (with eps 30 the first rung fails and the ladder is climbed)
        tp, wl, ws = int(rng.choice([0, 16, 64])), int(rng.choice([0, 2, 4, 8, 32])), int(rng.choice([64, 256]))
        ctx.set_option("tier_probe", tp)
        ctx.set_option("probe_min_pairs", 0.0 if tp else 5e10)
        ctx.set_option("wide_lists", wl)
        ctx.set_option("wide_sel", ws)
        # the bottom of the ladder: the exact kernels' two distance kernels, the shortcut to them, the ladder after the fact
        er, ed, el = int(rng.randint(0, 2)), int(rng.choice([0, 32, 4096])), int(rng.randint(0, 2))
        ctx.set_option("exact_rows", 2 if (er and ed == 32) else er)      # (2: + the one-pair-per-lane kernel for batches of >= 32 rows)
        ctx.set_option("esc_ladder", el)
        res = {}
        for prec in (0, 2, 1):   # fp16 first pass (default), split-bf16, float32 operands only
            ctx.set_option("precision", prec)
            try:
                ym = N.DeviceMatrix(ctx, t, metric)
                qm = ym if single else N.DeviceMatrix(ctx, s, metric)
                dd, ii, st = N.knn(ctx, qm, ym, k, exclude_self=single)
                res[prec] = (dd.numpy(), ii.numpy(), st)
            finally:
                ctx.set_option("precision", 0)
        ctx.set_option("short_ord_min_tiles", 48)
        ctx.set_option("eps_scale", 1.0)
        ctx.set_option("h_q64", 2)
        for name, v in (("tier_probe", 1024), ("probe_min_pairs", 5e10), ("wide_lists", 32), ("wide_sel", 256), ("exact_rows", 2),
                        ("esc_ladder", 1)):
            ctx.set_option(name, v)
        ok = all(np.array_equal(res[0][1], res[p][1]) and np.array_equal(res[0][0], res[p][0]) for p in (1, 2))
        if len(s) * n_t <= 2_000_000:
            od, oi = O.knn_exact(s, t, k, O.canonical_metric(metric), exclude_self=single)
            ok = ok and np.array_equal(res[0][1], oi)
        tag = "ok " if ok else "BAD"
        bad += 0 if ok else 1
        print(tag, f"n_s={len(s)} n_t={n_t} d={d} {metric} {np.dtype(dtype).name} k={k} single={single}",
              "tier", res[0][2]["first_pass"], "lists", res[0][2]["n_splits"], "x", res[0][2]["list_len"], f"min_tiles {min_tiles} eps {eps} q64 {q64} probe {tp} wide {wl}/{ws} -> {res[0][2]['wide_lists']} exact {er}/{ed} ladder {el}", "esc", res[0][2]["n_escalated_rows"], "fb", res[0][2]["n_fallback_rows"],
              "ratio %.3f" % res[0][2]["max_err_ratio"])
    print("cases", n_cases, "bad", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
