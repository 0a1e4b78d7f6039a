"""Time of the Minkowski-family metrics (exact float64 kernels) beside scikit-learn on the host cores.   python tools/minkowski_time.py [n] [d] [nosk]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kiez_amd import _native as N  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 300
with_sklearn = not (len(sys.argv) > 3 and sys.argv[3] == "nosk")     # (third argument "nosk": GPU times only)
ctx = N.Context.get()
rng = np.random.default_rng(0)
for dtype in (np.float32, np.float64):
    s, t = rng.standard_normal((n, d)).astype(dtype), rng.standard_normal((n, d)).astype(dtype)
    for mc in ("manhattan", "chebyshev", "minkowski[3.0]", "minkowski[1.5]", "euclidean"):
        sm, tm = N.DeviceMatrix(ctx, s, mc), N.DeviceMatrix(ctx, t, mc)
        N.knn(ctx, sm, tm, 10)
        ctx.sync()
        t0 = time.perf_counter()
        dd, ii, st = N.knn(ctx, sm, tm, 10)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3
        line = f"{dtype.__name__} {mc}: {ms:.1f} ms  ({n * n * d / ms / 1e6:.1f} G element-pairs/s)"
        if mc != "euclidean" and dtype == np.float32 and with_sklearn:
            from sklearn.neighbors import NearestNeighbors
            name, p = ("minkowski", float(mc[10:-1])) if mc.startswith("minkowski") else (mc, 2)
            nn = NearestNeighbors(n_neighbors=10, metric=name, p=p, algorithm="brute", n_jobs=-1).fit(t)
            t0 = time.perf_counter()
            di, ix = nn.kneighbors(s)
            sk_ms = (time.perf_counter() - t0) * 1e3
            bad = ix != ii.numpy()
            ties = (di[bad] == dd.numpy()[bad]).all()       # (same distance at every position whose index differs: exact ties in another order)
            line += f"   scikit-learn on the host cores: {sk_ms:.0f} ms; {int(bad.sum())} of {bad.size} indices differ, all of them exact ties: {bool(ties)}"
        print(line, flush=True)
