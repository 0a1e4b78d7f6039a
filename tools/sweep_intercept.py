"""GPU box: main-kernel time of the fp16 kernel against the LENGTH of the sweep (index rows) at a fixed number of query rows,
one index range per query tile: t = a + b * n_index; the intercept a is what the start of a sweep costs (ring prologue, list
initialisation and -- mostly -- the burst of events while the lists fill).   python3 tools/sweep_intercept.py [n_q] [d] [k]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N

n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = N.Context.get()
ctx.set_option("force_splits", 1)
rng = np.random.RandomState(0)
q = N.DeviceMatrix(ctx, rng.rand(n_q, d).astype(np.float32), "euclidean")
pts = []
for n_i in (12800, 25600, 51200, 102400, 204800, 409600):
    y = N.DeviceMatrix(ctx, rng.rand(n_i, d).astype(np.float32), "euclidean")
    ms = []
    for _ in range(6):
        _, _, st = N.knn(ctx, q, y, k)
        ms.append(st["main_kernel_ms"])
    ms = sorted(ms[1:])
    pts.append((n_i / 128, ms[len(ms) // 2]))
    print(f"n_index {n_i:7d} ({n_i // 128:5d} tiles): main kernel {ms[len(ms) // 2]:8.3f} ms  (min {ms[0]:.3f})  per tile {ms[len(ms) // 2] / (n_i / 128) * 1e3:7.3f} us", flush=True)
x, yv = np.array([p[0] for p in pts]), np.array([p[1] for p in pts])
b, a = np.polyfit(x[2:], yv[2:], 1)
print(f"fit over the four longest sweeps: t = {a:.3f} ms + {b * 1e3:.3f} us x tiles   (intercept = {a / yv[3] * 100:.1f} % of the {int(x[3])}-tile sweep)")
