"""GPU box: random cases through the three routes of the uncertified rows -- grouped ranges (default), one range per row (abl 16), the
whole index (exact_rows 2): neighbours and distance bits must be the same.  Shapes, metric, k, the kind of data (gaussian, tight
clusters, duplicated rows, a constant matrix) and an inflated rounding bound (so that every row fails every tier) are drawn at random.
    python3 tools/range_fuzz.py [cases] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402

ctx = N.Context.get()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def data(kind, n, d, r):
    if kind == "gauss":
        return r.standard_normal((n, d)).astype(np.float32)
    if kind == "tight":
        nc = int(r.integers(2, 12))
        c = r.standard_normal((nc, d)) * 3
        sc = 0.01 * 2.0 ** r.integers(0, 5, nc)
        a = r.integers(0, nc, n)
        return (c[a] + sc[a, None] * r.standard_normal((n, d))).astype(np.float32)
    if kind == "dups":
        base = r.standard_normal((max(n // 50, 3), d)).astype(np.float32)
        return base[r.integers(0, base.shape[0], n)]
    return np.full((n, d), 0.5, dtype=np.float32) + (r.random((n, d)) < 0.001).astype(np.float32)      # nearly constant


bad = n_range = n_grouped = 0
t0 = time.time()
for c in range(cases):
    d = int(rng.choice([16, 20, 32, 64, 100, 128, 200, 256, 260, 300, 512]))
    kind_next = str(rng.choice(["gauss", "tight", "tight", "dups", "const"]))
    n_q = int(rng.integers(150, 9000))
    if kind_next == "tight" and rng.random() < 0.5:
        n_q = int(rng.integers(6000, 24000))      # (enough uncertified rows for the grouped route)
    n_i = int(rng.integers(200, 20000))
    k = int(rng.choice([1, 5, 10, 50]))
    metric = str(rng.choice(["euclidean", "sqeuclidean", "cosine"]))
    kind = kind_next
    eps = float(rng.choice([1.0, 1.0, 30.0, 1e3, 1e30]))
    same = bool(rng.random() < 0.2)
    k = min(k, n_i - 1)
    y = data(kind, n_i, d, rng)
    q = y if same else data(kind, n_q, d, rng)
    if kind == "tight" and not same:      # (the same clusters on both sides)
        q = y[rng.integers(0, n_i, n_q)] + (1e-3 * rng.standard_normal((n_q, d))).astype(np.float32)
    ym = N.DeviceMatrix(ctx, y, metric)
    qm = ym if same else N.DeviceMatrix(ctx, q, metric)
    ctx.set_option("eps_scale", eps)
    out = []
    for er, abl in ((3, 0), (3, 16), (2, 0)):
        ctx.set_option("exact_rows", er)
        ctx.set_option("abl", abl)
        dd, ii, st = N.knn(ctx, qm, ym, k, exclude_self=same)
        out.append((dd.numpy(), ii.numpy(), st))
    ctx.set_option("exact_rows", 3)
    ctx.set_option("abl", 0)
    ctx.set_option("eps_scale", 1.0)
    ok = all(np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1]) for o in out[1:])
    st = out[0][2]
    n_range += st["n_range_rows"] > 0
    n_grouped += st["n_range_group_rows"] > 0
    if not ok:
        bad += 1
    if not ok or c % 20 == 0:
        print(f"case {c}: {'ok ' if ok else 'BAD'} {q.shape[0]} x {n_i} x {d} k={k} {metric} {kind} eps x{eps:g} self={same}: exact rows {st['n_fallback_rows']}"
              f" range {st['n_range_rows']} grouped {st['n_range_group_rows']} pairs {st['n_range_pairs']}; per row: range {out[1][2]['n_range_rows']}", flush=True)
    del qm, ym
print(f"{cases} cases ({n_range} with rows on the range re-search, {n_grouped} of them with groups), {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
