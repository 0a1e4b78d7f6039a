"""GPU box: cost per query row of the exact float64 kernels (every certification forced to fail: eps_scale = 1e30), 2 000 rows against
301 k index rows of d = 64, cosine, k = 50 and 10.      python3 tools/exact_prof.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N
ctx = N.Context.get()
rng = np.random.default_rng(0)
q = rng.standard_normal((2000, 64)).astype(np.float32); y = rng.standard_normal((301000, 64)).astype(np.float32)
qm, ym = N.DeviceMatrix(ctx, q, "cosine"), N.DeviceMatrix(ctx, y, "cosine")
ctx.set_option("eps_scale", 1e30)
for k in (50, 10):
    for _ in range(2):
        ctx.sync(); t0 = time.perf_counter()
        d, i, st = N.knn(ctx, qm, ym, k)
        ctx.sync(); ms = (time.perf_counter() - t0) * 1e3
    print("k", k, "ms", round(ms, 1), "us/row", round(ms * 1e3 / 2000, 1), "fallback rows", st["n_fallback_rows"], "fallback_ms", round(st["fallback_ms"], 1), flush=True)
