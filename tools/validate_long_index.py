import sys, time
import numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N
from oracle import kiez_oracle as O
rng = np.random.RandomState(0)
t = rng.rand(1_000_000, 300).astype(np.float32)
s = rng.rand(3000, 300).astype(np.float32)
ctx = N.Context.get()
ym = N.DeviceMatrix(ctx, t, "euclidean")
qm = N.DeviceMatrix(ctx, s, "euclidean")
t0 = time.time()
d, i, st = N.knn(ctx, qm, ym, 10)
print("gpu", time.time() - t0, st)
t0 = time.time()
od, oi = O.knn_exact(s, t, 10, "euclidean")
print("oracle", time.time() - t0)
print("rows identical", int((i.numpy() == oi).all(axis=1).sum()), "of", len(oi), "dist equal", bool(np.array_equal(d.numpy(), od)))
