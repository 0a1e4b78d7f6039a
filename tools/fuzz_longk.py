#!/usr/bin/env python3
"""Randomised check of kz_knn's long-k route (111 .. ~540 neighbours on the fused kernels) on the GPU box: random shapes,
metrics, dtypes, data kinds (incl. index rows in cluster order and exact duplicates), single-source mode and all three first-pass
tiers against the oracle's exact float64 search, bit for bit on the indices (distances: atol 1e-6 -- a row's distance to an
exact duplicate is exactly 0 here and sqrt(1e-14) from the oracle's expanded form).

    python3 tools/fuzz_longk.py [n_cases] [seed]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402
from oracle import kiez_oracle as O  # noqa: E402  (checker only)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = N.Context.get()
    bad = 0
    for c in range(n_cases):
        d = int(rng.choice([17, 24, 48, 64, 100, 128, 200, 300]))
        k = int(rng.integers(111, 541))
        n_t = int(rng.integers(max(4 * (k // 24 + 1) * 128, 3000), 60000))
        n_s = int(rng.choice([1, 130, 700, 2500]))
        metric = str(rng.choice(["euclidean", "sqeuclidean", "cosine"]))
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        kind = str(rng.choice(["uniform", "normal", "clustered", "dups"]))
        single = rng.random() < 0.2
        prec = int(rng.choice([0, 0, 0, 2, 1]))

        def gen(n):
            if kind == "uniform":
                return rng.random((n, d))
            if kind == "normal":
                return rng.standard_normal((n, d))
            if kind == "clustered":   # rows in cluster order: the nearest rows of a query are CONSECUTIVE index rows
                cen = rng.standard_normal((12, d)) * 4
                lab = np.sort(rng.integers(0, 12, n))
                return cen[lab] + 0.2 * rng.standard_normal((n, d))
            base = rng.random((max(n // 6, 4), d))
            return base[rng.integers(0, len(base), n)]
        t = gen(n_t).astype(dtype)
        s = t[: min(n_t, 3000)] if single else gen(n_s).astype(dtype)
        if single:
            t = s
        if metric == "cosine":
            s, t = s.astype(np.float64), (s if single else t).astype(np.float64)
        kk = min(k, len(t) - 2)
        ctx.set_option("precision", prec)
        ctx.set_option("short_ord_min_tiles", int(rng.choice([48, 1, 2, 4])))   # lists of 16 on the dealt image, forced onto small ranges
        try:
            ym = N.DeviceMatrix(ctx, t, metric)
            qm = ym if single else N.DeviceMatrix(ctx, s, metric)
            dd, ii, st = N.knn(ctx, qm, ym, kk, exclude_self=single)
        finally:
            ctx.set_option("precision", 0)
            ctx.set_option("short_ord_min_tiles", 48)
        od, oi = O.knn_exact(s, t, kk, O.canonical_metric(metric), exclude_self=single)
        ok = np.array_equal(ii.numpy(), oi) if kind != "dups" else np.allclose(dd.numpy(), od, rtol=1e-6, atol=1e-6)
        ok = ok and np.allclose(dd.numpy(), od, rtol=1e-6, atol=1e-6) and st["max_err_ratio"] < 1.0
        bad += 0 if ok else 1
        if not ok:
            dv, iv = dd.numpy(), ii.numpy()
            err = np.abs(dv - od)
            r, cc = np.unravel_index(np.argmax(err), err.shape)
            print(f"   max |dist - oracle| = {err.max():.3e} at row {r} col {cc}: {dv[r, cc]!r} vs {od[r, cc]!r}; index rows differing: {(iv != oi).any(axis=1).sum()}",
                  f"\n   row {r} dev idx {iv[r, max(0, cc - 3):cc + 4]} dist {dv[r, max(0, cc - 3):cc + 4]}\n   row {r} ora idx {oi[r, max(0, cc - 3):cc + 4]} dist {od[r, max(0, cc - 3):cc + 4]}", flush=True)
        print("ok " if ok else "BAD", f"n_s={len(s)} n_t={len(t)} d={d} {metric} {np.dtype(dtype).name} k={kk} {kind} single={single} prec={prec}",
              "lists", st["list_len"], "ranges", st["n_splits"], "esc", st["n_escalated_rows"], "fb", st["n_fallback_rows"], "ratio %.3f" % st["max_err_ratio"], flush=True)
    print("cases", n_cases, "bad", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
