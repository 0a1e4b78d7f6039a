timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_longk.py tests/test_gpu_short_lists.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^Extension" | tail -2
timeout 900 python3 tools/fuzz_longk.py 150 2307 2>&1 | tail -1
python3 - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from kiez_amd import _native as N
ctx = N.Context.get()
rng = np.random.RandomState(0)
q = rng.rand(50000, 200).astype(np.float32); y = rng.rand(500000, 200).astype(np.float32)
qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
for k in (128, 200, 320, 500):
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter()
        d, i, st = N.knn(ctx, qm, ym, k)
        ctx.sync(); ms = (time.perf_counter() - t0) * 1e3
    print(f"k={k}: call {ms:.1f} ms main {st['main_kernel_ms']:.1f} finalize {st['finalize_ms']:.1f} fallback {st['fallback_ms']:.1f} lists {st['n_splits']} x {st['list_len']} again {st['n_escalated_rows']}")
q = rng.rand(20000, 128).astype(np.float64); y = rng.rand(100000, 128).astype(np.float64)
qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
for k in (256, 512):
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter()
        d, i, st = N.knn(ctx, qm, ym, k)
        ctx.sync(); ms = (time.perf_counter() - t0) * 1e3
    print(f"float64 20k x 100k x 128 k={k}: call {ms:.1f} ms main {st['main_kernel_ms']:.1f} finalize {st['finalize_ms']:.1f}")
PY
