"""GPU box: the shared-sweep gate around its boundary -- kz_knn_dual as the library chooses, forced shared, forced two searches -- over
shapes on both sides of the gate (profiles/r06_gate_probe.log: chosen / best <= 1.07 everywhere).    python3 tools/gate_probe.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from kiez_amd import _native as N
ctx = N.Context.get()
rng = np.random.default_rng(1)
def best(fn, reps=4):
    b = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); fn(); ctx.sync()
        b = min(b, (time.perf_counter() - t0) * 1e3)
    return b
for n, d, k in [(30000,200,50),(40000,200,50),(50000,200,50),(60000,128,10),(70000,128,10),(80000,128,10),(90000,128,10),(30000,300,10),(40000,300,10),(50000,300,10),(40000,64,10),(80000,64,10),(120000,64,10)]:
    a, b = rng.random((n, d), dtype=np.float32), rng.random((n + 1000, d), dtype=np.float32)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    f = lambda: N.knn_dual(ctx, am, bm, k)
    ctx.set_option("dual_force", 0); ctx.set_option("dual_stride", 1)
    t_c = best(f); (_, _, sa), _ = N.knn_dual(ctx, am, bm, k)
    ctx.set_option("dual_force", 1); t_s = best(f)
    ctx.set_option("dual_force", 0); ctx.set_option("dual_stride", 0); t_t = best(f)
    ctx.set_option("dual_stride", 1)
    print(f"n {n:6d} d {d:3d} k {k:2d}: chosen {t_c:7.3f} (shared {sa['dual']})  shared {t_s:7.3f}  twice {t_t:7.3f}   chosen/best {t_c/min(t_s,t_t):.2f}", flush=True)
