"""GPU box: both directions of a fit (kz_knn_dual: the library chooses one shared sweep or two searches) over a ladder of sizes -- ms per
call, pairs per second, the route taken -- to see that throughput grows smoothly with the size (no shape at which a gate or a plan
boundary makes a larger problem cheaper than a smaller one).    python3 tools/shape_sweep.py [d=300] [k=10]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 300
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = N.Context.get()
rng = np.random.default_rng(0)
prev = None
for n in (2_000, 5_000, 10_000, 15_000, 20_000, 30_000, 40_000, 50_000, 60_000, 80_000, 100_000, 130_000, 160_000, 200_000, 300_000):
    a, b = rng.random((n, d), dtype=np.float32), rng.random((n + 1000, d), dtype=np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    best, st = 1e9, None
    for _ in range(4):
        ctx.sync()
        t0 = time.perf_counter()
        (_, _, sa), (_, _, sb) = N.knn_dual(ctx, am, bm, k)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    rate = 2.0 * n * (n + 1000) / best / 1e6
    flag = "" if prev is None or rate >= 0.9 * prev else "   <-- slower per pair than the smaller size"
    prev = rate
    print(f"n = {n:7d}  d = {d}  k = {k}: {best:9.3f} ms  {rate:9.1f} G pairs/s (both directions)  shared {sa['dual']}  ranges {sa['n_splits']}  "
          f"re-searched {sa['n_escalated_rows']}/{sb['n_escalated_rows']}  exact {sa['n_fallback_rows']}/{sb['n_fallback_rows']}{flag}", flush=True)
