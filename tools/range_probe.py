"""GPU box: the range re-search (kz_range.h) against the whole-index exact kernels on data whose tightest clusters no tier certifies
(tools/cliff_probe.py, last kind): same results, time, rows and pairs.      python3 tools/range_probe.py [n d k metric]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402

ctx = N.Context.get()


def gen(n, d, rng):
    centres = np.random.default_rng(5).standard_normal((40, d)) * 3
    sc = 0.05 * 2.0 ** np.random.default_rng(6).integers(0, 6, 40)
    c = rng.integers(0, 40, n)
    return (centres[c] + sc[c, None] * rng.standard_normal((n, d))).astype(np.float32)


SHAPES = ((30_000, 64, 10, "euclidean"), (60_000, 200, 50, "cosine"), (100_000, 128, 10, "euclidean"), (200_000, 200, 10, "euclidean"))
if len(sys.argv) > 4:
    SHAPES = ((int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]),)
for n, d, k, metric in SHAPES:
    rng = np.random.default_rng(11)
    a, b = gen(n, d, rng), gen(n + 1000, d, rng)
    am, bm = N.DeviceMatrix(ctx, a, metric), N.DeviceMatrix(ctx, b, metric)
    res = {}
    for er in ((int(os.environ['ONLY']),) if os.environ.get('ONLY') else (3, 2)):
        ctx.set_option("exact_rows", er)
        best = None
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            dd, ii, st = N.knn(ctx, am, bm, k)
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3
            best = ms if best is None or ms < best else best
        res[er] = (dd.numpy(), ii.numpy())
        print(f"{n} x {n + 1000} x {d} k={k} {metric} exact_rows={er}: {best:8.1f} ms  fallback_ms {st['fallback_ms']:8.1f}  exact rows {st['n_fallback_rows']}"
              f"  range rows {st['n_range_rows']} (grouped {st['n_range_group_rows']})  pairs {st['n_range_pairs']}  re-searched {st['n_escalated_rows']}", flush=True)
    if len(res) < 2:
        continue
    same_i = np.array_equal(res[3][1], res[2][1])
    same_d = np.array_equal(res[3][0], res[2][0])
    print(f"    same neighbours {same_i}  same distances {same_d}", flush=True)
    if not same_i:
        bad = np.nonzero((res[3][1] != res[2][1]).any(axis=1))[0]
        print("    rows that differ:", len(bad), bad[:10], flush=True)
    del am, bm
ctx.set_option("exact_rows", 3)
