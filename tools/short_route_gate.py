"""GPU box: where the short-list route of kz_knn_dual starts to pay -- 500k x nb, d = 200, k = 50, uniform data, route forced on
(dual_short_min_tiles = 1) and off, for several nb (tiles per index range = nb / 128 / 10).  python3 tools/short_route_gate.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N
ctx = N.Context.get()
rng = np.random.default_rng(9)
d, k = 200, 50
a = rng.random((500000, d)).astype(np.float32)
am = N.DeviceMatrix(ctx, a, "cosine")
for nb in (62500, 83000, 100000, 125000, 170000, 250000):
    b = rng.random((nb, d)).astype(np.float32)
    bm = N.DeviceMatrix(ctx, b, "cosine")
    out = []
    for on in (0, 1, 0, 1):
        ctx.set_option("dual_short_main", on)
        ctx.set_option("dual_short_min_tiles", 1)
        ctx.sync()
        t0 = time.perf_counter()
        (xd, xi, sa), (yd, yi, sb) = N.knn_dual(ctx, am, bm, k)
        ctx.sync()
        out.append(((time.perf_counter() - t0) * 1e3, sa["main_kernel_ms"], sa["n_splits"], sa["n_escalated_rows"]))
    print(f"nb {nb:7d} ({nb / 128 / 10:5.1f} tiles per range): off {out[2][0]:6.1f} ms (main {out[2][1]:5.1f})   on {out[3][0]:6.1f} ms (main {out[3][1]:5.1f}, "
          f"{out[3][2]} ranges, {out[3][3]} rows again)", flush=True)
ctx.set_option("dual_short_main", 1)
ctx.set_option("dual_short_min_tiles", 128)
