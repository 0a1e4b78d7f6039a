"""GPU box: randomised shapes for kz_knn_dual (shared sweep forced) against two kz_knn calls, bit for bit.
   python3 tools/fuzz_dual.py [n_cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
_o = sys.argv[3] if len(sys.argv) > 3 else "-1"          # run just this case, or the cases "first-last" (the random stream is advanced through the others)
only_first, only = (int(_o.split("-")[0]), int(_o.split("-")[1])) if "-" in _o[1:] else (int(_o), int(_o))
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 1     # multiplies the row counts (10: up to 600 k x 400 k)
ctx = N.Context.get()
bad = 0
for case in range(n_cases):
    na = int(rng.integers(1100, 60000)) * scale
    nb = int(rng.integers(1100, 40000)) * scale
    d = int(rng.choice([17, 24, 32, 40, 48, 64, 72, 100, 128, 129, 200, 208, 256, 300, 384]))
    k = int(rng.choice([1, 2, 3, 5, 10, 12, 13, 20, 26, 27, 50, 54, 60, 100, 110]))
    k = min(k, na, nb)
    metric = str(rng.choice(["euclidean", "sqeuclidean", "cosine"]))
    dtype = np.float32 if rng.random() < 0.7 else np.float64
    kind = str(rng.choice(["uniform", "normal", "clustered", "dups", "scaled"]))
    def gen(n):
        if kind == "uniform":
            return rng.random((n, d))
        if kind == "normal":
            return rng.standard_normal((n, d))
        if kind == "scaled":
            return rng.standard_normal((n, d)) * 1e3 + 5e3
        if kind == "clustered":
            c = rng.standard_normal((6, d)) * 4
            return c[rng.integers(0, 6, n)] + 0.3 * rng.standard_normal((n, d)) * rng.random((n, 1)) * 3
        base = rng.random((max(n // 5, 4), d))
        return base[rng.integers(0, len(base), n)]
    a, b = gen(na).astype(dtype), gen(nb).astype(dtype)
    opt = (int(rng.choice([1, 1, 2, 4, 8, 16, 32])), int(rng.choice([0, 0, 4096, 16384])), int(rng.integers(0, 2)),
           int(rng.integers(0, 2)), int(rng.integers(0, 2)),   # ..., wide workgroups for the shared sweep, reverse chain on the second stream
           int(rng.choice([128, 3, 3, 8])), int(rng.integers(0, 2)),   # short-list route: smallest index range in tiles; short sample lists
           int(rng.integers(0, 2)), int(rng.choice([16, 48, 48, 100])), int(rng.integers(0, 2)))   # reverse lists of 2 K'; entries selected beyond k; bf16 tier in the chain
    q64 = int(rng.integers(0, 3))             # 64-queries-per-wave kernel: 0 nowhere, 1 in the reference searches only, 2 in the shared sweep only
    if os.environ.get("KZ_FUZZ_Q64") is not None:
        q64 = int(os.environ["KZ_FUZZ_Q64"])
    fl = (int(rng.integers(0, 2)), int(rng.choice([64, 256, 2048])), float(rng.choice([0.3, 1.3, 1.3, 3.0])))   # seeded lists: on / probe rows / margin
    nested = int(rng.integers(0, 3) > 0)      # nested sample (the sampled rows are swept by the sample sweep only): on in two cases of three
    if os.environ.get("KZ_FUZZ_NESTED") is not None:
        nested = int(os.environ["KZ_FUZZ_NESTED"])
    if os.environ.get("KZ_FUZZ_FLOOR") is not None:
        fl = (int(os.environ["KZ_FUZZ_FLOOR"]),) + fl[1:]
    if only >= 0 and not (only_first <= case <= only):
        continue
    print(f"case {case}: na={na} nb={nb} d={d} k={k} {metric} {dtype.__name__} {kind} stride/chunk/deal/wide/overlap/short-min/short-sample/rev-long/extra/bf {opt}", flush=True)
    ctx.set_option("dual_stride", opt[0])
    ctx.set_option("chunk_rows", opt[1])
    am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
    ctx.set_option("dual_force", 0)
    ctx.set_option("list_floor", 0)
    ctx.set_option("h_q64", 1 if q64 == 1 else 0)
    r1 = N.knn(ctx, am, bm, k); ctx.sync()
    if only >= 0: print("  a->b done", r1[2]["n_escalated_rows"], r1[2]["n_fallback_rows"], flush=True)
    r2 = N.knn(ctx, bm, am, k); ctx.sync()
    if only >= 0: print("  b->a done", r2[2]["n_escalated_rows"], r2[2]["n_fallback_rows"], flush=True)
    ctx.set_option("dual_force", 1)
    ctx.set_option("dual_overlap", opt[4])
    ctx.set_option("dual_short_min_tiles", opt[5])
    ctx.set_option("short_ord_min_tiles", min(opt[5], 48))
    ctx.set_option("dual_sample_short", opt[6])
    ctx.set_option("dual_rev_long", opt[7])
    ctx.set_option("dual_short_extra", opt[8])
    ctx.set_option("esc_bf", opt[9])
    ctx.set_option("h_q64", 1 if q64 == 2 else 0)
    ctx.set_option("list_floor", fl[0])
    ctx.set_option("floor_margin", fl[2])
    ctx.set_option("dual_nested", nested)
    (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, k)
    ctx.set_option("dual_nested", 1)
    ctx.set_option("list_floor", 0)
    ctx.set_option("h_q64", 2)
    ok = (np.array_equal(r1[1].numpy(), xi.numpy()) and np.array_equal(r1[0].numpy(), xd.numpy())
          and np.array_equal(r2[1].numpy(), yi.numpy()) and np.array_equal(r2[0].numpy(), yd.numpy()))
    if not ok:
        # diagnosis of a failing case: the reference searches once more on the other kernel, the oracle on the differing rows
        from oracle import kiez_oracle as O
        ctx.set_option("dual_force", 0)
        ctx.set_option("h_q64", 0 if q64 == 1 else 1)
        o1 = N.knn(ctx, am, bm, k); o2 = N.knn(ctx, bm, am, k)
        ctx.set_option("h_q64", 2)
        for name, ref, alt, dual_i, qa, ya in (("a->b", r1, o1, xi, a, b), ("b->a", r2, o2, yi, b, a)):
            ri, ai, di = ref[1].numpy(), alt[1].numpy(), dual_i.numpy()
            rows = np.nonzero((ri != di).any(axis=1))[0]
            print(f"  {name}: rows where reference != dual: {len(rows)}; reference(other kernel) != dual: {int((ai != di).any(axis=1).sum())}; "
                  f"reference != reference(other kernel): {int((ri != ai).any(axis=1).sum())}", flush=True)
            if len(rows):
                mc = O.canonical_metric(metric)
                q64_ = qa[rows[:8]].astype(np.float64) if mc == "cosine" else qa[rows[:8]]
                y64_ = ya.astype(np.float64) if mc == "cosine" else ya
                od, oi = O.knn_exact(q64_, y64_, k, mc)
                for j, r in enumerate(rows[:8]):
                    print(f"    row {r}: oracle==reference {np.array_equal(oi[j], ri[r])} oracle==dual {np.array_equal(oi[j], di[r])} "
                          f"first diff col {int(np.nonzero(ri[r] != di[r])[0][0])} ref {ri[r][:6]} dual {di[r][:6]} oracle {oi[j][:6]} dist {od[j][:4]}", flush=True)
    ratio = max(sa["max_err_ratio"], sb["max_err_ratio"])
    if not ok or ratio >= 1.0:
        bad += 1
    print(("ok " if ok else "BAD"), f"na={na} nb={nb} d={d} k={k} {metric} {dtype.__name__} {kind} dual {sa['dual']}/{sb['dual']} "
          f"q64 {q64} floor {fl} nested {nested} ev/row {sb['n_events'] / nb:.1f} esc {sa['n_escalated_rows']}/{sb['n_escalated_rows']} splits {sa['n_splits']} ovf {sb['n_overflow_rows']} ratio {ratio:.3f}", flush=True)
for name, v in (("dual_force", 0), ("dual_stride", 1), ("chunk_rows", 0), ("dual_overlap", 1),
                ("dual_short_min_tiles", 128), ("short_ord_min_tiles", 48), ("dual_sample_short", 1), ("dual_rev_long", 1), ("dual_short_extra", 48), ("esc_bf", 1), ("list_floor", 1), ("floor_margin", 1.3)):
    ctx.set_option(name, v)
print("cases", n_cases, "bad", bad)
sys.exit(1 if bad else 0)
