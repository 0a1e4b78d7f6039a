python3 - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N
ctx = N.Context.get()
rng = np.random.RandomState(0)
q = rng.rand(20000, 128).astype(np.float32); y = rng.rand(100000, 128).astype(np.float32)
qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
for k in (128, 256, 512):
    res = {}
    for lk in (1, 0):
        ctx.set_option("long_k", lk)
        ts = []
        for _ in range(3):
            ctx.sync(); t0 = time.perf_counter(); d, i, st = N.knn(ctx, qm, ym, k); ctx.sync(); ts.append(time.perf_counter() - t0)
        res[lk] = (min(ts), i.numpy(), st)
    assert np.array_equal(res[0][1], res[1][1])
    print(f"20k x 100k x 128, k={k}: long-k route {res[1][0]*1e3:.1f} ms (ranges {res[1][2]['n_splits']}, fallback rows {res[1][2]['n_fallback_rows']}), exact kernels {res[0][0]*1e3:.1f} ms; identical indices")
ctx.set_option("long_k", 1)
PY
