"""The hand-written stable radix sort of (float key, int value) pairs (csrc/kz_sort.hip: orders rows by their event threshold in the
shared sweep; no counterpart in the reference) against numpy's stable sort -- both directions, ties, signed zeros, infinities, sizes
around the tile of 2 048 pairs.  `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_SYMBOL = "_Z21kz_sort_pairs_f32_i32P6kz_ctxPKfPfPKiPiii"   # (internal C++ entry point of the library, kz_common.h)


def _sort(ctx, keys, vals, descending):
    from kiez_amd import _native as N
    fn = getattr(ctx.lib, _SYMBOL)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int]
    n = len(keys)
    dk, dv = ctx.to_device(keys), ctx.to_device(vals)
    ok, ov = ctx.empty((max(n, 1),), np.float32), ctx.empty((max(n, 1),), np.int32)
    N._check(fn(ctx.handle, dk.ptr, ok.ptr, dv.ptr, ov.ptr, n, int(descending)), "kz_sort_pairs_f32_i32")
    ctx.sync()
    return ok.numpy()[:n], ov.numpy()[:n]


@pytest.mark.parametrize("n", [1, 2, 63, 64, 255, 256, 257, 2047, 2048, 2049, 4097, 100_000, 1_000_003])
@pytest.mark.parametrize("descending", [0, 1])
def test_stable_sort_matches_numpy(n, descending):
    from kiez_amd import _native as N
    ctx = N.Context.get()
    rng = np.random.default_rng(n + descending)
    kind = n % 3
    if kind == 0:
        keys = rng.standard_normal(n).astype(np.float32) * np.float32(10.0 ** rng.integers(-30, 30))
    elif kind == 1:
        keys = rng.integers(-3, 4, n).astype(np.float32)            # many ties: stability decides the order of the values
    else:
        keys = rng.standard_normal(n).astype(np.float32)
        keys[rng.integers(0, n, max(n // 10, 1))] = np.float32(np.inf)
        keys[rng.integers(0, n, max(n // 10, 1))] = np.float32(-np.inf)
        keys[rng.integers(0, n, max(n // 10, 1))] = np.float32(0.0)
        keys[rng.integers(0, n, max(n // 10, 1))] = np.float32(-0.0)
    vals = rng.permutation(n).astype(np.int32)
    gk, gv = _sort(ctx, keys, vals, descending)
    # the order of the bit patterns: -0.0 sorts below +0.0 (as rocPRIM's radix sort did); stable in both directions
    bits = keys.view(np.uint32)
    sortable = np.where(bits >> 31, ~bits, bits | np.uint32(0x80000000)).astype(np.uint32)
    if descending:
        sortable = ~sortable
    order = np.argsort(sortable, kind="stable")
    np.testing.assert_array_equal(gv, vals[order])
    np.testing.assert_array_equal(gk.view(np.uint32), keys[order].view(np.uint32))
