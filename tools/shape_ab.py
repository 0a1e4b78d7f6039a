"""Diagnostic: main-kernel time of kz_knn for one shape with the library named by KIEZ_AMD_LIB (ablation builds give WRONG
results -- timing only).  python tools/shape_ab.py n_q n_i d k [option=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kiez_amd import _native as N
n_q, n_i, d, k = (int(x) for x in sys.argv[1:5])
ctx = N.Context.get()
for o in sys.argv[5:]:      # context options name=value
    name, _, val = o.partition("=")
    ctx.set_option(name, float(val))
rng = np.random.RandomState(0)
q = rng.rand(n_q, d).astype(np.float32)
y = rng.rand(n_i, d).astype(np.float32)
qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
ms = []
for _ in range(4):
    _, _, st = N.knn(ctx, qm, ym, k)
    ms.append(st["main_kernel_ms"])
print(os.path.basename(os.environ.get("KIEZ_AMD_LIB", "default")), sys.argv[1:], "main_kernel_ms", " ".join("%.3f" % m for m in ms), "esc", st["n_escalated_rows"])
